#!/bin/bash
# speculation hints that follow a monotone run of the strong-Wolfe search: GPU tier, then strict-Wolfe workloads and the
# configs that must not move
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_mom
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
one() { local tag=$1; shift
    timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline > $OUT/o.json 2> $OUT/o.err || { echo "$tag failed"; tail -2 $OUT/o.err; return; }
    python3 -c "
import json; d=json.loads(open('$OUT/o.json').read().strip().splitlines()[-1]); print('$tag value %.0f med %.0f it/s trials/iter %.2f launches/iter %.2f' % (d['value'], d['value_median'], d['config']['trials_per_iteration'], d['config']['launches_per_iteration']), {k:(v['launches'], round(v['avg_us'],1)) for k,v in d['kernels'].items()})"
}
for rep in 1 2; do
one c1 --workload c1 --steps 15 --warmup 3 --windows 1
one c1c --workload c1c --steps 15 --warmup 3 --windows 1
done
one rosenPR_1e6 --workload c1 --size 1e6 --steps 15 --warmup 3 --windows 1
one rosenPR_1e7 --workload c1 --size 1e7 --steps 15 --warmup 3 --windows 1
one c3 --workload c3 --steps 200 --warmup 10 --windows 2
one c2 --workload c2 --steps 300 --warmup 10 --windows 3
one c5 --steps 20 --warmup 5 --windows 3
