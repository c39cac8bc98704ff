#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_lb
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
for n in 1e5 1e6; do
CGO_BENCH_NO_PROFILE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$n -- python3 $R/bench.py --workload c2 --beta LBFGS --size $n --steps 200 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/t_$n.log 2>&1; echo "rc=$?"; tail -1 $OUT/t_$n.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LBFGS n=$n', round(d['value'],1), d['config']['launches_per_iteration'], d['config']['trials_per_iteration'])"
(cd $R && python3 scripts/gap_table.py $OUT/t_$n --skip 60 2>&1 | tail -22 | cut -c1-150)
done
find $OUT -name '*kernel_trace.csv' -size +5M -delete
