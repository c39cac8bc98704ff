#!/bin/bash
# The pure-HBM threshold (1.4 GB per read-write launch) was measured when the contiguous chunks were not line-aligned: the
# quadratic (seven points, 40 B/element) by size, library policy (grid-stride below 1.4 GB = n 3.5e7) against contiguous chunks +
# non-temporal accesses forced (CGO_BIG_BYTES=1), no placement search, alternating.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OUT=$R/gpurun_out/r04_bigthr; mkdir -p $OUT
for rep in 1 2; do for n in 1.25e7 1.6e7 2e7 2.5e7 3e7 3.4e7; do for big in lib forced; do
  if [ $big = forced ]; then export CGO_BIG_BYTES=1; else unset CGO_BIG_BYTES; fi
  timeout -k 10 200 python3 bench.py --size $n --steps 60 --warmup 10 --windows 2 --no-cpu-baseline --no-placement-search > $OUT/q_${n}_${big}_$rep.json 2> $OUT/q_${n}_${big}_$rep.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/q_${n}_${big}_$rep.json").read().strip().splitlines()[-1])
print(f"n=$n $big rep=$rep: {d['value']:8.0f} [{d['value_median']:8.0f}] it/s  {d['roofline']['kernel']} {d['roofline']['avg_launch_us']:6.1f} us frac {d['roofline']['frac']:.3f}")
PY
done; done; done
