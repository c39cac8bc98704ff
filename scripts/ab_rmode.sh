#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "=== R-mode (default)"; python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5', round(d['value'],1),'it/s trials/iter',round(d['config']['trials_per_iteration'],2),'launches/iter',round(d['config']['launches_per_iteration'],2), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'])) for k,v in d['kernels'].items()})"
echo "=== stored-g"; CGO_STORED_G=1 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c5', round(d['value'],1),'it/s trials/iter',round(d['config']['trials_per_iteration'],2),'launches/iter',round(d['config']['launches_per_iteration'],2), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'])) for k,v in d['kernels'].items()})"
for w in "c2 --steps 300" "c3 --steps 200"; do
python3 bench.py --workload $w --warmup 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:28], round(d['value'],1),'it/s trials/iter',round(d['config']['trials_per_iteration'],2),'launches/iter',round(d['config']['launches_per_iteration'],2), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'])) for k,v in d['kernels'].items()})"
done
python3 scripts/latency_check.py 2>&1 | grep -v "^ "
