#!/bin/bash
# Grid / streaming-policy sweep for the shard sizes of the N = 8 / 4 / 2 runs (n = 1e8 / N), 5- and 7-point launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; k=d["kernels"]["accept_dir_trial"]; print(os.environ.get("TAG",""), c["n"], d["kernel_family"][22:29], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2), "ADT", round(k["avg_us"],1), "us", round(k["gbps"]), "GB/s kernel frac", round(d["kernel_time_fraction_of_wall"],3))'
for n in 12500000 25000000 50000000; do
  for m7 in 10000000 9000000000000000000; do
    for g in 256 512 1024 2048; do
      TAG="grid=$g" CGO_GRID_SMALL=$g CGO_BIG_BYTES=1e12 CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c2 --size $n --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="grid=$g" python3 -c "$fmt"
    done
    TAG="BIG(4096,NT)" CGO_BIG_BYTES=1 CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c2 --size $n --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="BIG" python3 -c "$fmt"
    TAG="default" CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c2 --size $n --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="default" python3 -c "$fmt"
  done
done
