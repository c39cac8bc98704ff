#!/bin/bash
# A/B of the on-device controller (CGO_CTL_DEPTH) on the four bench workloads + two small sizes.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(c["workload"][:30], round(d["value"],1),"it/s trials/it",round(c["trials_per_iteration"],2),"launches/it",round(c["launches_per_iteration"],2),"ctl/it",round(c["controller_armed_launches_per_iteration"],2), {k:(v["launches"],round(v["avg_us"],1)) for k,v in d["kernels"].items()})'
for depth in 0 8; do
  echo "=== CGO_CTL_DEPTH=$depth"
  for w in "c2 --steps 400" "c2 --size 100000 --steps 400" "c3 --steps 200" "c3 --size 1000000 --steps 400" "c5 --steps 60" "c2 --beta HagerZhang --steps 400"; do
    CGO_CTL_DEPTH=$depth python3 bench.py --workload $w --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "$fmt"
  done
done
