#!/bin/bash
# n = 1.25e7: what the launch costs beyond its bare stream mix — reduction tail on / off, 1 / 3 / 7 points, and the bare mix itself
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_shard
mkdir -p $OUT
cd $R
BIGN=9000000000000000000
run() { tag=$1; shift; env CGO_PLACE_TUNE=0 "$@" python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/t_$tag.json 2> $OUT/t_$tag.err
  echo "== $tag: $(python3 -c "import json; d=json.load(open('$OUT/t_$tag.json')); k=d['kernels']; print(round(d['value']), round(d['value_median']), 'it/s;', {n: (v['launches'], round(v['avg_us'],1)) for n,v in k.items()}, 'launches/it', round(d['config']['launches_per_iteration'],2))")"; }
run p7_fused
run p7_unfused CGO_FUSED_TAIL=0
run p1_fused CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN
run p1_unfused CGO_FUSED_TAIL=0 CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN
run p7_g256 CGO_GRID_CG7=256
python3 - <<'PY'
import cgo_amd as cgo
ctx = cgo.default_context()
print("bare mix (BIG policy) at n=1.25e7: median/best us", cgo.bench_stream_mix(12500000, 15, ctx))
PY
