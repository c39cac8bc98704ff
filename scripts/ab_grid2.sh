#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; k=d["kernels"]["accept_dir_trial"]; print(os.environ.get("TAG",""), c["n"], d["kernel_family"][22:29], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2), "ADT", round(k["avg_us"],1), "us", round(k["gbps"]), "GB/s kernel frac", round(d["kernel_time_fraction_of_wall"],3))'
for n in 100000000; do
  for g in 256 512 1024; do
    TAG="grid=$g" CGO_GRID_SMALL=$g CGO_BIG_BYTES=1e12 python3 bench.py --workload c5 --size $n --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | TAG="grid=$g" python3 -c "$fmt"
  done
  for rep in 1 2; do
  TAG="default" python3 bench.py --workload c5 --size $n --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | TAG="default(BIG)" python3 -c "$fmt"
  done
done
for n in 12500000 25000000; do for g in 384 512 768; do
    TAG="grid=$g" CGO_GRID_SMALL=$g CGO_BIG_BYTES=1e12 CGO_MULTI7_MIN_N=10000000 python3 bench.py --workload c2 --size $n --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="grid=$g" python3 -c "$fmt"
done; done
