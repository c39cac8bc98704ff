#!/bin/bash
# Config 4 (L-BFGS m = 10 on log-sum-exp, n = 1e7) in its two-pass form (CGO_LBFGS_SPEC=0): the push forming g⁺ itself
# (k_lbfgs_push_gram_lse) vs materialize() + plain push (CGO_LBFGS_FUSE_GRAD=0), same box, alternating.  Output: gpurun_out/r03_c4g/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_c4g
mkdir -p $OUT
cd $R
for rep in 1 2; do
for f in 1 0; do
  CGO_LBFGS_SPEC=0 CGO_LBFGS_FUSE_GRAD=$f python3 bench.py --workload c4 --steps 45 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c4_g${f}_$rep.json 2> $OUT/c4_g${f}_$rep.err
  echo "fuse_grad=$f rep=$rep rc=$?"
  python3 - <<PY
import json
d = json.load(open("$OUT/c4_g${f}_$rep.json"))
print("  it/s first %.1f median %.1f | launches/it %.2f |" % (d["value"], d["value_median"], d["config"]["launches_per_iteration"]),
      {k: (v["launches"], round(v["avg_us"], 1)) for k, v in d["kernels"].items()})
PY
done
done
