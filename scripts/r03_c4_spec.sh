#!/bin/bash
# Config 4 (L-BFGS m = 10 on log-sum-exp, n = 1e7): ONE ring pass per iteration — the state update riding in the next direction pass
# (k_lbfgs_combine_spec<…, PUSH>; default, here =2) or as its own launch (CGO_LBFGS_SPEC=1: + k_lbfgs_push_lite) — vs the
# two-pass form (CGO_LBFGS_SPEC=0: k_lbfgs_push_gram_lse + k_lbfgs_combine_lse), same box, alternating.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_c4s
mkdir -p $OUT
cd $R
for rep in 1 2; do
for f in 2 1 0; do
  CGO_LBFGS_SPEC=$f python3 bench.py --workload c4 --steps 45 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/c4_s${f}_$rep.json 2> $OUT/c4_s${f}_$rep.err
  echo "spec=$f rep=$rep rc=$?"
  python3 - <<PY
import json
d = json.load(open("$OUT/c4_s${f}_$rep.json"))
print("  it/s first %.1f median %.1f | launches/it %.2f |" % (d["value"], d["value_median"], d["config"]["launches_per_iteration"]),
      {k: (v["launches"], round(v["avg_us"], 1)) for k, v in d["kernels"].items()})
PY
done
done
