#!/bin/bash
# n = 1.25e7 (the 8-GPU shard): grid size and trial points per launch of the grid-stride 7-point launch, after the DPP tail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_shard
mkdir -p $OUT
cd $R
BIGN=9000000000000000000
run() { tag=$1; shift; env CGO_PLACE_TUNE=0 "$@" python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/g_$tag.json 2> $OUT/g_$tag.err
  echo "== $tag: $(python3 -c "import json; d=json.load(open('$OUT/g_$tag.json')); k=d['kernels']['accept_dir_trial']; print(round(d['value']), round(d['value_median']), 'it/s; accept_dir_trial', round(k['avg_us'],1), 'us; launches/it', round(d['config']['launches_per_iteration'],2), d['roofline']['kernel'])")"; }
for g in 512 768 1024 1536 2048 4096; do run p7_g$g CGO_GRID_CG7=$g; done
for g in 512 1024 2048; do run p5_g$g CGO_GRID_CG7=$g CGO_MULTI7_MIN_N=$BIGN; done
for g in 256 512 1024; do run p3_g$g CGO_GRID_SMALL=$g CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN; done
run p7_big CGO_BIG_BYTES=4e8
