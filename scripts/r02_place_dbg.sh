#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CGO_DEBUG_PLACE=1 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>&1 >/dev/null | grep "cgo place" | head
CGO_DEBUG_PLACE=1 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['accept_dir_trial']['avg_us'],1), round(d['value'],1), round(d['value_median'],1), d['roofline'], d.get('placement'))"
