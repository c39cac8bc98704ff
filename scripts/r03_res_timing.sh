#!/bin/bash
# In-kernel phase timing of resident slices (CGO_RES_TIMING=1): configs 1 and 2, 1 / 3 / 7 points, a few chunk sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_restime
mkdir -p $OUT
cd $R
for w in c1 c2; do
  steps=200; [ $w = c1 ] && steps=15
  for pts in 3 1 7; do
    CGO_RES_TIMING=1 CGO_RES_POINTS=$pts CGO_BENCH_NO_PROFILE=1 python3 bench.py --workload $w --steps $steps --warmup 3 --windows 3 --no-cpu-baseline > $OUT/${w}_p$pts.json 2> $OUT/${w}_p$pts.err
    echo "== $w pts=$pts: $(cat $OUT/${w}_p$pts.json | cut -c1-200)"; grep "cgo resident" $OUT/${w}_p$pts.err | tail -2
  done
done
for chunk in 1024 2048 6000; do
    CGO_RES_TIMING=1 CGO_RES_CHUNK=$chunk CGO_BENCH_NO_PROFILE=1 python3 bench.py --workload c2 --steps 200 --warmup 3 --windows 3 --no-cpu-baseline --size 2e5 > $OUT/c2s_c$chunk.json 2> $OUT/c2s_c$chunk.err
    echo "== n=2e5 chunk=$chunk: $(cat $OUT/c2s_c$chunk.json | cut -c1-120)"; grep "cgo resident" $OUT/c2s_c$chunk.err | tail -1
done
