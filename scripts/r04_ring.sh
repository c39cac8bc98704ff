#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_ring
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_baseline_sizes.py tests/test_policy.py tests/test_long_horizon.py -m gpu -q -x -k "lbfgs or lse or config4 or LBFGS or c4" > $OUT/tests.log 2>&1; rc=$?; echo "lbfgs tests rc=$rc"; tail -3 $OUT/tests.log
[ $rc = 0 ] || exit $rc
for rep in 1 2; do for n in 1e7 10000001 10000008; do
  timeout -k 10 200 python3 bench.py --workload c4 --size $n --steps 45 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/c4_${n}_$rep.json 2> $OUT/c4_${n}_$rep.err; echo "c4 n=$n rep=$rep rc=$? $(python3 -c "
import json;d=json.loads(open('$OUT/c4_${n}_$rep.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['value_median']),d['roofline']['kernel'],round(d['roofline']['avg_launch_us'],1),round(d['roofline']['frac'],3))")"
done; done
