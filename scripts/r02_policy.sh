#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_policy
mkdir -p $OUT
cd $R
python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -n "^FAILED\|^ERROR\|^E  " $OUT/pytest_gpu.log | head -30; exit 1; }
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2>$OUT/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','value_median','value_min','value_max','ms_per_step')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['value'])"
show() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1])
except Exception as e:
    print('$2 FAILED', open('$1'.replace('.json','.err')).read()[-400:]); sys.exit(0)
print('$2', 'value %.0f med %.0f it/s'%(d['value'],d['value_median']), 'ctl/iter %.2f trials/iter %.2f'%(d['controller_armed_launches_per_iteration'], d['trials_per_iteration']))
"; }
export CGO_BENCH_NO_PROFILE=1
for n in 1e4 1e5 1e6 3e6 1.25e7; do
  for g in 128 256 512 1024; do
    CGO_GRID_CG7=$g python3 bench.py --workload c2 --size $n --steps 300 --warmup 20 --windows 3 > $OUT/q_${n}_g$g.json 2>$OUT/q_${n}_g$g.err; show $OUT/q_${n}_g$g.json "quad n=$n 7pt grid7=$g"
  done
done
