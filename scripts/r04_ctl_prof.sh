#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_ctl_prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
export CGO_CTL_DEPTH=4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/armed -- python3 $R/bench.py --size 1.25e7 --steps 40 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/armed.log 2>&1; echo "armed rc=$?"
f=$(find $OUT/armed -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-200
t=$(find $OUT/armed -name '*kernel_trace.csv' | head -1)
python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-40:]
prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(r["Kernel_Name"][:60], "dur %.1f us"%((e-s)/1e3), "gap %.1f us"%(((s-prev)/1e3) if prev else 0), "scratch", r.get("Scratch_Size") or r.get("Private_Segment_Size"), "grid", r.get("Grid_Size"))
    prev=e
PY
