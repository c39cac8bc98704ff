#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for size in 1.25e7 2.5e7 5e7; do for big in 1e8 1e12; do
CGO_BIG_BYTES=$big python3 bench.py --size $size --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n=$size big_bytes=$big', round(d['value'],1),'it/s', {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'])) for k,v in d['kernels'].items()})"
done; done
