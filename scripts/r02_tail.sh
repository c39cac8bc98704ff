#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_tail
mkdir -p $OUT
cd $R
python3 -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && { grep -n "^FAILED\|^ERROR\|^E  " $OUT/pytest_gpu.log | head -30; exit 1; }
show() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1])
except Exception as e:
    print('$2 FAILED', open('$1'.replace('.json','.err')).read()[-400:]); sys.exit(0)
if 'kernels' in d:
    print('$2', 'value %.1f med %.1f it/s'%(d['value'],d['value_median']), 'trials/iter %.2f launches/iter %.2f'%(d['config']['trials_per_iteration'], d['config']['launches_per_iteration']), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'] or 0)) for k,v in d['kernels'].items()})
else:
    print('$2', 'value %.0f med %.0f it/s'%(d['value'],d['value_median']), 'ctl/iter %.2f trials/iter %.2f'%(d['controller_armed_launches_per_iteration'], d['trials_per_iteration']))
"; }
python3 bench.py --steps 50 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/c5.json 2>$OUT/c5.err; show $OUT/c5.json "c5"
for g in 512 1024 2048; do
  CGO_GRID_CG7=$g python3 bench.py --size 1.25e7 --steps 200 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/shard_g$g.json 2>$OUT/shard_g$g.err; show $OUT/shard_g$g.json "shard 1.25e7 grid7=$g"
done
export CGO_BENCH_NO_PROFILE=1
P7="CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=0 CGO_MULTI7_MIN_N=0"
for n in 1e4 1e5 1e6 3e6 1.25e7; do
  for cfg in "0 1" "8 0" "8 1"; do set -- $cfg
    env $P7 CGO_CTL_DEPTH=$1 CGO_CTL_GRAPH=$2 python3 bench.py --workload c2 --size $n --steps 300 --warmup 20 --windows 3 > $OUT/q_${n}_d$1_g$2.json 2>$OUT/q_${n}_d$1_g$2.err; show $OUT/q_${n}_d$1_g$2.json "quad n=$n 7pt depth=$1 graph=$2"
  done
done
python3 bench.py --workload c2 --steps 300 --warmup 20 --windows 3 > $OUT/c2_default.json 2>$OUT/c2_default.err; show $OUT/c2_default.json "c2 default policy"
for cfg in "0 1" "4 0" "4 1"; do set -- $cfg
  CGO_CTL_DEPTH=$1 CGO_CTL_GRAPH=$2 python3 bench.py --workload c3 --steps 200 --warmup 10 --windows 2 > $OUT/c3_d$1_g$2.json 2>$OUT/c3_d$1_g$2.err; show $OUT/c3_d$1_g$2.json "c3 depth=$1 graph=$2"
  CGO_CTL_DEPTH=$1 CGO_CTL_GRAPH=$2 python3 bench.py --workload c3 --size 1e6 --steps 200 --warmup 10 --windows 2 > $OUT/c3s_d$1_g$2.json 2>$OUT/c3s_d$1_g$2.err; show $OUT/c3s_d$1_g$2.json "c3 n=1e6 depth=$1 graph=$2"
done
for p in 3 5 7; do
  m5=9000000000000000000; m7=9000000000000000000; [ $p -ge 5 ] && m5=0; [ $p -ge 7 ] && m7=0
  CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=$m5 CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c3 --steps 200 --warmup 10 --windows 2 > $OUT/c3_p$p.json 2>$OUT/c3_p$p.err; show $OUT/c3_p$p.json "c3 points=$p"
done
