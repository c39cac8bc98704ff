#!/bin/bash
# default launch policy across sizes + Rosenbrock 1 vs 3 points at small n
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; k=d["kernels"]["accept_dir_trial"]; print(os.environ.get("TAG",""), c["workload"][:22], c["n"], d["kernel_family"][22:29], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2), "ADT", round(k["avg_us"],1), "us kernel frac", round(d["kernel_time_fraction_of_wall"],3))'
for n in 10000 100000 1000000 3000000 10000000; do
  TAG="default" python3 bench.py --workload c2 --size $n --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="default" python3 -c "$fmt"
done
BIGN=9000000000000000000
for n in 100000 1000000 3000000; do
  for m in $BIGN 0; do
    TAG="rosen MULTI_MIN_N=$m" CGO_MULTI_MIN_N=$m python3 bench.py --workload c3 --size $n --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="rosen min_n=$m" python3 -c "$fmt"
  done
done
