#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
fmt='import json,sys,os; d=json.loads(sys.stdin.read()); c=d["config"]; print(os.environ.get("TAG",""), c["workload"][:28], c["n"], round(d["value"],1),"it/s launches/it",round(c["launches_per_iteration"],2),"ctl/it",round(c["controller_armed_launches_per_iteration"],2), {k:(v["launches"],round(v["avg_us"],1)) for k,v in d["kernels"].items()})'
for w in "c3 --size 1000000 --steps 400" "c3 --steps 300" "c3 --size 100000 --steps 400"; do
  for depth in 0 2 4 8; do
    TAG="depth=$depth" CGO_CTL_DEPTH=$depth python3 bench.py --workload $w --warmup 20 --no-cpu-baseline 2>/dev/null | TAG="depth=$depth" python3 -c "$fmt"
  done
done
cd /tmp && CGO_CTL_DEPTH=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ctl -- python3 $R/bench.py --workload c3 --size 1000000 --steps 400 --warmup 20 --no-cpu-baseline > $R/gpurun_out/prof_ctl.log 2>&1
f=$(ls -t $R/gpurun_out/prof_ctl/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-170
