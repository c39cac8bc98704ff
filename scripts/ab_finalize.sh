#!/bin/bash
# A/B of the two-stage finalize threshold on mid-size problems (256 rows of 40–56 slots).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(c["workload"][:30], d["kernel_family"], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2), "kernel frac", round(d["kernel_time_fraction_of_wall"],3))'
for thr in 65536 131072 262144; do
  echo "=== CGO_FINALIZE_2STAGE_BYTES=$thr"
  for w in "c2 --size 10000000 --steps 300" "c2 --size 12500000 --steps 300" "c2 --size 16000000 --steps 300"; do
    for m7 in 10000000 9000000000000000000; do
      CGO_MULTI7_MIN_N=$m7 CGO_FINALIZE_2STAGE_BYTES=$thr python3 bench.py --workload $w --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$fmt"
    done
  done
done
