#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_c1p
mkdir -p $OUT
cd $R
for rep in 1 2 3; do
for d in 0 4; do
    CGO_CTL_DEPTH=$d timeout -k 10 200 python3 bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/o.json 2> $OUT/o.err || { echo "failed"; tail -2 $OUT/o.err; continue; }
    python3 -c "
import json; d=json.loads(open('$OUT/o.json').read().strip().splitlines()[-1]); print('c1 depth=$d value %.0f it/s launches/iter %.2f armed/iter %s' % (d['value'], d['config']['launches_per_iteration'], d['config']['controller_armed_launches_per_iteration']), {k:(v['launches'], round(v['avg_us'],1)) for k,v in d['kernels'].items()}, round(d['kernel_time_fraction_of_wall'],3))"
done
done
