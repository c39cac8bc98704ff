#!/bin/bash
# 3-point stencil launches: full GPU tier, then config 1's chained form with one and three trial points per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_chain3
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
BIG=9000000000000000000
for rep in 1 2; do
for m in $BIG 0; do
    CGO_MULTI_MIN_N=$m timeout -k 10 300 python3 bench.py --workload c1c --steps 20 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/c1c_m$m.json 2> $OUT/c1c_m$m.err || { echo failed; tail -3 $OUT/c1c_m$m.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/c1c_m$m.json').read().strip().splitlines()[-1]); print('c1c multi_min_n=$m value %.0f it/s launches/iter %s trials/iter %s' % (d['value'], d.get('launches_per_iteration'), d.get('trials_per_iteration')), d['roofline']['kernel'])"
    CGO_MULTI_MIN_N=$m timeout -k 10 300 python3 bench.py --workload c1c --size 1e6 --beta HagerZhang --steps 100 --warmup 5 --windows 2 --no-cpu-baseline > $OUT/c1c6_m$m.json 2> $OUT/c1c6_m$m.err || { echo failed; tail -3 $OUT/c1c6_m$m.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/c1c6_m$m.json').read().strip().splitlines()[-1]); print('chain n=1e6 HZ multi_min_n=$m value %.0f it/s launches/iter %s' % (d['value'], d.get('launches_per_iteration')), d['roofline']['kernel'], d.get('stopped_early'))"
done
done
