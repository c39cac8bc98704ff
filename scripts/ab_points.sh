#!/bin/bash
# A/B: 3-point vs 5-point speculative launches (CGO_MULTI5_MIN_N) on the bench workloads.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(c["workload"][:30], d["kernel_family"], round(d["value"],1),"it/s trials/it",round(c["trials_per_iteration"],2),"launches/it",round(c["launches_per_iteration"],2), {k:(v["launches"],round(v["avg_us"],1),round(v["gbps"])) for k,v in d["kernels"].items()})'
for cfg in "9000000000000000000 9000000000000000000" "0 9000000000000000000" "0 0"; do
  set -- $cfg; m5=$1; export CGO_MULTI7_MIN_N=$2
  echo "=== CGO_MULTI5_MIN_N=$m5 CGO_MULTI7_MIN_N=$2"
  for w in "c5 --steps 100" "c3 --steps 200" "c2 --size 10000000 --steps 200" "c5 --size 30000000 --steps 100"; do
    CGO_MULTI5_MIN_N=$m5 python3 bench.py --workload $w --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "$fmt"
  done
done
