#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_grid7
mkdir -p $OUT
cd $R
export CGO_BENCH_NO_PROFILE=1
for n in 5e5 1e6 2e6 3e6; do
  for g in 128 192 256 384 512; do
    CGO_GRID_CG7=$g timeout -k 10 300 python3 bench.py --size $n --steps 300 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/q_${n}_g$g.json 2> $OUT/q_${n}_g$g.err || { echo failed; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/q_${n}_g$g.json').read().strip().splitlines()[-1]); print('n=$n grid=$g value %.0f med %.0f it/s' % (d['value'], d['value_median']))"
  done
done
