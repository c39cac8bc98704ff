#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; k=d["kernels"]["accept_dir_trial"]; print(os.environ.get("TAG",""), c["workload"][:24], d["kernel_family"][22:29], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2), "ADT", round(k["avg_us"],1), "us", round(k["gbps"]), "GB/s kernel frac", round(d["kernel_time_fraction_of_wall"],3))'
for g in 256 512 1024; do
  TAG="grid=$g" CGO_GRID_SMALL=$g python3 bench.py --workload c3 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="grid=$g" python3 -c "$fmt"
  TAG="grid=$g" CGO_GRID_SMALL=$g CGO_MULTI5_MIN_N=9000000000000000000 python3 bench.py --workload c2 --size 10000000 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="grid=$g 3pt" python3 -c "$fmt"
done
for w in "c2 --size 12500000 --steps 200" "c2 --size 25000000 --steps 200" "c5 --steps 60"; do
  TAG="default" python3 bench.py --workload $w --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="default" python3 -c "$fmt"
done
