#!/bin/bash
# Round-1 profile collection on the GPU box: bench line, rocprofv3 kernel stats, PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $R && python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err && tail -c 600 $OUT/bench_n1.json && \
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_stats.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/prof_write.log 2>&1
echo "exit=$?"
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -type f | head -30
