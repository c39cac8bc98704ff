#!/bin/bash
# one finalize launch (k_finalize_one) against two (k_finalize_t x 2) behind the pure-HBM launches of the headline config
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_fin1
mkdir -p $OUT
cd $R
for v in 1 0 1 0; do
    CGO_FUSED_TAIL=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --windows 5 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err || { echo failed; tail -3 $OUT/c5.err; }
    python3 -c "
import json; d=json.loads(open('$OUT/c5.json').read().strip().splitlines()[-1]); print('fused_tail=$v value %.1f med %.1f max %.1f kernel %.1f us frac_wall %.3f' % (d['value'], d['value_median'], d['value_max'], d['roofline']['avg_launch_us'], d['kernel_time_fraction_of_wall']), d['placement'])"
done
