#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_restime
mkdir -p $OUT
cd $R
for w in c1 c2; do
  steps=200; [ $w = c1 ] && steps=15
  CGO_RES_TIMING=1 CGO_BENCH_NO_PROFILE=1 python3 bench.py --workload $w --steps $steps --warmup 3 --windows 3 --no-cpu-baseline > $OUT/${w}_t2.json 2> $OUT/${w}_t2.err
  echo "== $w: $(cat $OUT/${w}_t2.json | cut -c1-100)"; grep "cgo resident" $OUT/${w}_t2.err | tail -2
done
