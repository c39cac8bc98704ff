#!/bin/bash
# After the armed launches stopped spilling their argument copy to scratch memory: the on-device controller (4 rounds deep)
# against host-driven launches over sizes, for the cheap class (quadratic, seven points) and extended Rosenbrock (three points).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_ctl_sweep
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "controller or armed or ctl or pipe" > $OUT/tests.log 2>&1; echo "controller tests rc=$?"; tail -3 $OUT/tests.log
export CGO_RESIDENT=0
run() { # tag workload size depth
  CGO_CTL_DEPTH=$4 timeout -k 10 200 python3 bench.py --workload $2 --size $3 --steps 100 --warmup 10 --windows 3 --no-cpu-baseline --no-placement-search > $OUT/$1_$3_d$4.json 2> $OUT/$1_$3_d$4.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/$1_$3_d$4.json").read().strip().splitlines()[-1])
print("$1 n=$3 depth=$4: it/s", round(d["value"]), "median", round(d.get("value_median") or 0), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],1), "launches/iter", d["config"]["launches_per_iteration"], "armed/iter", d["config"]["controller_armed_launches_per_iteration"])
PY
}
for n in 1e5 1e6 3e6 1.25e7 3e7; do for d in 0 4; do run quad c5 $n $d; done; done
for n in 1e5 1e6 3e6 1e7 4e7; do for d in 0 4; do run rosen c3 $n $d; done; done
for n in 1.25e7; do for d in 0 2 8; do run quad c5 $n $d; done; done
