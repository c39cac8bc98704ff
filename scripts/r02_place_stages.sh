#!/bin/bash
# placement search with up to three stages of eight spare buffers: how often does a process end on a fast triple?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_stages
mkdir -p $OUT
cd $R
for rep in 1 2 3 4 5 6 7 8; do
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err || { echo failed; tail -3 $OUT/c5.err; }
    python3 -c "
import json; d=json.loads(open('$OUT/c5.json').read().strip().splitlines()[-1]); print('run $rep value %.1f med %.1f kernel %.1f us' % (d['value'], d['value_median'], d['roofline']['avg_launch_us']), d['placement'])"
done
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "2_to_31 or eight_billion or full_size or streaming_path or handles" 2>&1 | tail -3
