#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_c4d
mkdir -p $OUT
cd $R
for v in "A=1" "CGO_TAIL_STRICT=1" "CGO_FUSED_TAIL=0" "A=2"; do
    env $v timeout -k 10 300 python3 bench.py --workload c4 --steps 45 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c4.json 2> $OUT/c4.err || { echo failed; tail -3 $OUT/c4.err; }
    python3 -c "
import json; d=json.loads(open('$OUT/c4.json').read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['value_median'],1), round(d['value_max'],1), round(d['kernel_time_fraction_of_wall'],3))"
done
