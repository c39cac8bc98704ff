#!/bin/bash
# bench.py --gpus N with N ranks on ONE GPU (gloo rendezvous; the mailbox carries the exchange, RCCL refuses several ranks on one
# device and is skipped): the multi-rank code path of the final build, not a scaling number.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_rehn
mkdir -p $OUT
cd $R
for N in 2 4; do
    timeout -k 10 500 python3 bench.py --gpus $N --backend gloo --size 4e7 --steps 30 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/rehearse$N.json 2> $OUT/rehearse$N.err; echo "N=$N rc=$?"
    python3 -c "
import json; d=json.loads(open('$OUT/rehearse$N.json').read().strip().splitlines()[-1]); print(d['n_gpus'], round(d['value'],1), d['config'].get('comm'), {k:(round(v['value'],1), v['n_ranks_seen'], v['exchange_wait_us_per_launch']) for k,v in d['transports'].items()}, d.get('transports_failed'))"
    grep -v "^\[W\|Gloo\|amdgpu.ids" $OUT/rehearse$N.err | tail -3
done
