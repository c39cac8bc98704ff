#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in "c2 --steps 600" "c2 --size 1500000 --steps 600" "c3 --size 1000000 --steps 600" "c3 --steps 300"; do
  for depth in 0 4; do
    echo -n "noprof $w depth=$depth: "; CGO_BENCH_NO_PROFILE=1 CGO_CTL_DEPTH=$depth python3 bench.py --workload $w --warmup 20 --no-cpu-baseline 2>/dev/null
  done
done
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; print(c["workload"][:28], c["n"], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2),"ctl/it",round(c["controller_armed_launches_per_iteration"],2), {k:(v["launches"],round(v["avg_us"],1)) for k,v in d["kernels"].items()})'
for w in "c2 --steps 600" "c3 --steps 300" "c3 --size 1000000 --steps 600" "c4 --steps 100"; do
  python3 bench.py --workload $w --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "$fmt"
done
