#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2 3 4; do python3 bench.py --steps 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps50', round(d['value'],1), {k:(v['launches'],round(v['avg_us'],1)) for k,v in d['kernels'].items()})"; done
for i in 1 2; do python3 bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps20', round(d['value'],1), {k:(v['launches'],round(v['avg_us'],1)) for k,v in d['kernels'].items()})"; done
for i in 1 2; do python3 bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps200', round(d['value'],1), {k:(v['launches'],round(v['avg_us'],1)) for k,v in d['kernels'].items()})"; done
