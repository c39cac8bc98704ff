#!/bin/bash
# Where does a resident slice spend its time?  Kernel trace of configs 1 and 2, noinline vs inline evaluator builds.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_resprof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for lib in libcgo_hip.so libcgo_hip_inl.so; do
  for w in c1 c2; do
    steps=200; [ $w = c1 ] && steps=15
    tag=${w}_${lib%.so}
    CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python3 $R/bench.py --workload $w --steps $steps --warmup 3 --windows 2 --no-cpu-baseline > $OUT/$tag.log 2>&1
    echo "== $tag rc=$?"; grep -h '^{' $OUT/$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'it/s', d['config']['launches_per_iteration'], d['kernels'].get('resident'))"
    f=$(find $OUT/$tag -name '*kernel_stats.csv' | head -1); head -4 $f | cut -c1-200
  done
done
find $OUT -name '*kernel_trace.csv' -size +3M -delete
