#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_misc
mkdir -p $OUT
cd $R
python3 scripts/debug_closure.py > $OUT/debug_closure.log 2>&1; cat $OUT/debug_closure.log | tail -30
show() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1])
except Exception as e:
    print('$2 FAILED', open('$1'.replace('.json','.err')).read()[-300:]); sys.exit(0)
if 'kernels' in d:
    print('$2', 'value %.1f med %.1f it/s'%(d['value'],d['value_median']), 'trials/iter %.2f launches/iter %.2f'%(d['config']['trials_per_iteration'], d['config']['launches_per_iteration']), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'] or 0)) for k,v in d['kernels'].items()})
else:
    print('$2', d)
"; }
# same box: the R/W-mix harness next to the engine, in place and ping-pong
timeout -k 10 200 scripts/tune/rw_mix 1e8 9 > $OUT/rw_mix_1e8.log 2>&1; grep -n "in place.*chunk/WG    U2 thr256  ntL ntS  grid= 4096\|out of place" $OUT/rw_mix_1e8.log
python3 bench.py --steps 50 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/c5_inplace.json 2>$OUT/c5_inplace.err; show $OUT/c5_inplace.json "c5 in-place"
CGO_PINGPONG=1 python3 bench.py --steps 50 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/c5_pp.json 2>$OUT/c5_pp.err; show $OUT/c5_pp.json "c5 pingpong"
# n = 1e6: points x grid
export CGO_BENCH_NO_PROFILE=1
for p in 3 7; do for g in 128 256 512; do
  m5=9000000000000000000; m7=9000000000000000000; [ $p -ge 5 ] && m5=0; [ $p -ge 7 ] && m7=0
  CGO_GRID_SMALL=$g CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=$m5 CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c2 --steps 300 --warmup 10 --windows 3 > $OUT/c2_p${p}_g$g.json 2>$OUT/c2_p${p}_g$g.err; show $OUT/c2_p${p}_g$g.json "c2 n=1e6 points=$p grid7=$g"
done; done
for d in 0 4 8; do
  CGO_CTL_DEPTH=$d CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=0 CGO_MULTI7_MIN_N=0 python3 bench.py --workload c2 --steps 300 --warmup 10 --windows 3 > $OUT/c2_p7_ctl$d.json 2>$OUT/c2_p7_ctl$d.err; show $OUT/c2_p7_ctl$d.json "c2 n=1e6 points=7 ctl=$d"
done
unset CGO_BENCH_NO_PROFILE
for w in "c3 200 2" "c4 45 2"; do set -- $w
  python3 bench.py --workload $1 --steps $2 --warmup 10 --windows $3 > $OUT/$1.json 2>$OUT/$1.err; show $OUT/$1.json "$1"
done
for p in 3 5 7; do
  m5=9000000000000000000; m7=9000000000000000000; [ $p -ge 5 ] && m5=0; [ $p -ge 7 ] && m7=0
  CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=$m5 CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c3 --steps 200 --warmup 10 --windows 2 > $OUT/c3_p$p.json 2>$OUT/c3_p$p.err; show $OUT/c3_p$p.json "c3 points=$p"
done
