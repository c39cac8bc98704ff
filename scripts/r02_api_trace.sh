#!/bin/bash
# Where does the host turnaround of a latency-bound launch go?  HIP API trace (no counters) of BASELINE config 2.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_api
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CGO_BENCH_NO_PROFILE=1 rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/bench.py --workload c2 --steps 300 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/t.log 2>&1; echo "rc=$?"
f=$(ls $OUT/t/*/*hip_api_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -15 "$f"
find $OUT -name '*_trace.csv' -size +20M -delete
ls $OUT/t/*/ | head
