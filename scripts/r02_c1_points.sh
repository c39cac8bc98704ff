#!/bin/bash
# BASELINE config 1 (paired Rosenbrock n = 1000, PR-CG, strong Wolfe c2 = 0.1: ~7 trial steps per line search): trial points per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_c1p
mkdir -p $OUT
cd $R
BIG=9000000000000000000
for pts in 3 5 7; do
  case $pts in 3) M="$BIG $BIG";; 5) M="0 $BIG";; 7) M="0 0";; esac
  set -- $M
  for d in 0 4; do
    for w in c1 c1c; do
    CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=$1 CGO_MULTI7_MIN_N=$2 CGO_CTL_DEPTH=$d timeout -k 10 200 python3 bench.py --workload $w --steps 15 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/o.json 2> $OUT/o.err || { echo "failed"; tail -2 $OUT/o.err; continue; }
    python3 -c "
import json; d=json.loads(open('$OUT/o.json').read().strip().splitlines()[-1]); print('$w points=$pts depth=$d value %.0f it/s trials/iter %.2f launches/iter %.2f' % (d['value'], d['config']['trials_per_iteration'], d['config']['launches_per_iteration']), d['roofline']['kernel'])"
    done
  done
done
