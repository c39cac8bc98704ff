#!/bin/bash
# Round 4, second batch on the GPU box: the named log-sum-exp regression cases, the barrier-method diagnostic, 5 vs 7 trial
# points at the 8-GPU shard size, the long-horizon records on the final build.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_b2
mkdir -p $OUT
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "speculated_sums_are_not_used" > $OUT/regress.log 2>&1; echo "regression rc=$?"; tail -3 $OUT/regress.log
timeout -k 10 300 python3 scripts/r04_barrier_diag.py > $OUT/barrier_diag.log 2>&1; echo "barrier diag rc=$?"; tail -5 $OUT/barrier_diag.log
W="--size 1.25e7 --steps 100 --warmup 10 --windows 5 --no-cpu-baseline"
for rep in 1 2; do
  for pts in 7 5 3; do
    if [ $pts = 7 ]; then E=""; elif [ $pts = 5 ]; then E="CGO_MULTI7_MIN_N=1000000000000"; else E="CGO_MULTI7_MIN_N=1000000000000 CGO_MULTI5_MIN_N=1000000000000"; fi
    env $E timeout -k 10 200 python3 bench.py $W > $OUT/shard_p${pts}_$rep.json 2> $OUT/shard_p${pts}_$rep.err; echo "shard points=$pts rep=$rep rc=$?"
    python3 - <<PY
import json
d=json.loads(open("$OUT/shard_p${pts}_$rep.json").read().strip().splitlines()[-1])
print("  it/s", d["value"], "median", d.get("value_median"), "kernel", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], "launches/iter", d["config"].get("launches_per_iteration"), "trials/iter", d["config"].get("trials_per_iteration"))
PY
  done
done
CGO_LONG_HORIZON_OUT=$OUT timeout -k 10 600 python3 -m pytest tests/test_long_horizon.py -m gpu -q > $OUT/long_horizon.log 2>&1; echo "long horizon rc=$?"; tail -3 $OUT/long_horizon.log
