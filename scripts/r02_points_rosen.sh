#!/bin/bash
# Trial points per launch for the non-cheap class (extended Rosenbrock), again, with the cheap reductions of round 2:
# 1 point (policy below n = 3e6) vs 3 vs 7, host-driven and controller-armed, events off.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_pr
mkdir -p $OUT
cd $R
export CGO_BENCH_NO_PROFILE=1
BIG=9000000000000000000
for n in 1e3 1e4 1e5 1e6 3e6; do
  for pts in 1 3 7; do
    case $pts in 1) M="$BIG $BIG $BIG";; 3) M="0 $BIG $BIG";; 7) M="0 0 0";; esac
    set -- $M
    for d in 0 4; do
      CGO_MULTI_MIN_N=$1 CGO_MULTI5_MIN_N=$2 CGO_MULTI7_MIN_N=$3 CGO_CTL_DEPTH=$d timeout -k 10 300 python3 bench.py --workload c3 --size $n --steps 200 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/r_${n}_p${pts}_d$d.json 2> $OUT/r_${n}_p${pts}_d$d.err || { echo failed; tail -3 $OUT/r_${n}_p${pts}_d$d.err; exit 1; }
      python3 -c "
import json; d=json.loads(open('$OUT/r_${n}_p${pts}_d$d.json').read().strip().splitlines()[-1]); print('rosen HZ n=$n points=$pts depth=$d value %.0f med %.0f it/s launches/iter %.2f' % (d['value'], d['value_median'], d.get('launches_per_iteration') or 0))"
    done
  done
done
for pts in 1 3 7; do
    case $pts in 1) M="$BIG $BIG $BIG";; 3) M="0 $BIG $BIG";; 7) M="0 0 0";; esac
    set -- $M
    CGO_MULTI_MIN_N=$1 CGO_MULTI5_MIN_N=$2 CGO_MULTI7_MIN_N=$3 timeout -k 10 300 python3 bench.py --workload c1 --steps 200 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c1_p$pts.json 2> $OUT/c1_p$pts.err
    python3 -c "
import json; d=json.loads(open('$OUT/c1_p$pts.json').read().strip().splitlines()[-1]); print('c1 (rosen n=1000 PR SW) points=$pts value %.0f med %.0f it/s launches/iter %.2f' % (d['value'], d['value_median'], d.get('launches_per_iteration') or 0), d.get('stopped_early'))"
done
