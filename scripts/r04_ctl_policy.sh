#!/bin/bash
# Library policy check for the on-device controller (extended Rosenbrock, three points, HZ + Wolfe; resident solver off, no
# HIP-event ring): host-driven against 4 armed rounds in flight, by size.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_ctl_policy
mkdir -p $OUT
cd $R
export CGO_RESIDENT=0 CGO_BENCH_NO_PROFILE=1
for rep in 1 2; do
for n in 1e4 1e5 3e5 1e6 3e6; do for d in 0 4; do
  CGO_CTL_DEPTH=$d timeout -k 10 200 python3 bench.py --workload c3 --size $n --steps 200 --warmup 20 --windows 5 --no-cpu-baseline --no-placement-search > $OUT/rosen_${n}_d${d}_$rep.json 2> $OUT/rosen_${n}_d${d}_$rep.err
  echo "rosen n=$n depth=$d rep=$rep: $(tail -1 $OUT/rosen_${n}_d${d}_$rep.json | cut -c1-220)"
done; done; done
