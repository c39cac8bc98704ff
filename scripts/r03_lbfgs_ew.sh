#!/bin/bash
# L-BFGS m = 10 on the separable quadratic (n = 1e7, 3e7) and on paired Rosenbrock (n = 1e7): one ring pass per iteration
# (default) vs the two-pass form with its trial launch (CGO_LBFGS_SPEC=0), same box, alternating.  Output gpurun_out/r03_lbew/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_lbew
mkdir -p $OUT
cd $R
for rep in 1 2; do
for f in 2 0; do
  for cfg in "c5 1e7" "c5 3e7" "c3 1e7"; do
    set -- $cfg
    CGO_LBFGS_SPEC=$f python3 bench.py --workload $1 --size $2 --beta LBFGS --steps 40 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/$1_$2_s${f}_$rep.json 2> $OUT/$1_$2_s${f}_$rep.err
    python3 - <<PY
import json
try:
    d = json.load(open("$OUT/$1_$2_s${f}_$rep.json"))
    print("spec=$f $1 n=$2: it/s first %.1f median %.1f | trials/it %.2f launches/it %.2f |" % (d["value"], d["value_median"], d["config"]["trials_per_iteration"], d["config"]["launches_per_iteration"]),
          {k: (v["launches"], round(v["avg_us"], 1)) for k, v in d["kernels"].items()})
except Exception as e:
    print("spec=$f $1 n=$2: failed", e, open("$OUT/$1_$2_s${f}_$rep.err").read()[-300:])
PY
  done
done
done
