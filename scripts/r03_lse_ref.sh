#!/bin/bash
# k_lse_stats with the fixed reference (default) vs the running maximum (CGO_LSE_REF=0): log-sum-exp n = 1e7 under PR-CG
# (bench.py --workload c4 --beta PolakRibiere: the trial kernel is most of the iteration) and under L-BFGS two-pass, same box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_lseref
mkdir -p $OUT
cd $R
for rep in 1 2; do
for f in 1 0; do
  CGO_LSE_REF=$f python3 bench.py --workload c4 --beta PolakRibiere --steps 60 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/pr_r${f}_$rep.json 2> $OUT/pr_r${f}_$rep.err
  python3 - <<PY
import json
try:
    d = json.load(open("$OUT/pr_r${f}_$rep.json"))
    print("ref=$f rep=$rep PR-CG: it/s %.1f [%.1f] trials/it %.2f launches/it %.2f |" % (d["value"], d["value_median"], d["config"]["trials_per_iteration"], d["config"]["launches_per_iteration"]),
          {k: (v["launches"], round(v["avg_us"], 1)) for k, v in d["kernels"].items()})
except Exception as e:
    print("ref=$f failed", e, open("$OUT/pr_r${f}_$rep.err").read()[-300:])
PY
done
done
