#!/bin/bash
# Is the controller still worth arming anywhere once host-driven launches finish their own sums?  Rosenbrock / HZ at the
# sizes where round 1 measured +10-14 %, host-driven (depth 0) against armed rounds (depth 4), events off.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_cp
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "controller or fused" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && exit 1
export CGO_BENCH_NO_PROFILE=1
for n in 1e4 1e5 1e6 3e6 1e7; do
  for d in 0 4 0 4; do
    CGO_CTL_DEPTH=$d timeout -k 10 300 python3 bench.py --workload c3 --size $n --steps 200 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c3_${n}_d$d.json 2> $OUT/c3_${n}_d$d.err || { echo failed; tail -3 $OUT/c3_${n}_d$d.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/c3_${n}_d$d.json').read().strip().splitlines()[-1]); print('c3 n=$n depth=$d value %.0f med %.0f it/s' % (d['value'], d['value_median']), d.get('stopped_early'))"
  done
done
for d in 0 4; do
    CGO_CTL_DEPTH=$d timeout -k 10 300 python3 bench.py --workload c2 --steps 300 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c2_d$d.json 2> $OUT/c2_d$d.err
    python3 -c "
import json; d=json.loads(open('$OUT/c2_d$d.json').read().strip().splitlines()[-1]); print('c2 depth=$d value %.0f med %.0f it/s' % (d['value'], d['value_median']))"
done
echo done
