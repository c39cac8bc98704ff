#!/bin/bash
# workgroup cap of the 1/3-point grid-stride launches (extended Rosenbrock, HZ + weak Wolfe), events off
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_grid3
mkdir -p $OUT
cd $R
export CGO_BENCH_NO_PROFILE=1
for n in 1e6 3e6 1e7 2e7; do
  for g in 256 512 1024; do
    CGO_GRID_SMALL=$g timeout -k 10 300 python3 bench.py --workload c3 --size $n --steps 200 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/r_${n}_g$g.json 2> $OUT/r_${n}_g$g.err || { echo failed; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/r_${n}_g$g.json').read().strip().splitlines()[-1]); print('rosen n=$n grid=$g value %.0f med %.0f it/s' % (d['value'], d['value_median']))"
  done
done
