#!/bin/bash
# The on-device controller at the 8-GPU shard size (n = 1.25e7, seven trial points per launch): host-driven (library policy)
# against armed rounds 2 / 4 deep, alternating on one box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_ctl
mkdir -p $OUT
cd $R
W="--size 1.25e7 --steps 100 --warmup 10 --windows 5 --no-cpu-baseline"
for rep in 1 2; do
  for d in 0 2 4; do
    CGO_CTL_DEPTH=$d timeout -k 10 200 python3 bench.py $W > $OUT/d${d}_$rep.json 2> $OUT/d${d}_$rep.err; echo "depth $d rep=$rep rc=$?"
    python3 - <<PY
import json
d=json.loads(open("$OUT/d${d}_$rep.json").read().strip().splitlines()[-1])
print("  it/s", round(d["value"]), "median", round(d.get("value_median") or 0), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],1), "launches/iter", d["config"]["launches_per_iteration"], "armed/iter", d["config"]["controller_armed_launches_per_iteration"])
PY
  done
done
