#!/bin/bash
# Multi-point launches at small n (Infinity-Cache resident): is a saved launch worth the wider reduction?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; k=d["kernels"]["accept_dir_trial"]; print(os.environ.get("TAG",""), c["n"], d["kernel_family"][22:29], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2), "ADT", round(k["avg_us"],1), "us kernel frac", round(d["kernel_time_fraction_of_wall"],3))'
BIGN=9000000000000000000
for n in 10000 100000 1000000 3000000; do
  for cfg in "$BIGN $BIGN $BIGN" "0 $BIGN $BIGN" "0 0 $BIGN" "0 0 0"; do
    set -- $cfg
    CGO_MULTI_MIN_N=$1 CGO_MULTI5_MIN_N=$2 CGO_MULTI7_MIN_N=$3 python3 bench.py --workload c2 --size $n --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="" python3 -c "$fmt"
  done
done
