#!/bin/bash
# bench.py with two ranks on ONE GPU (gloo rendezvous, mailbox transport; RCCL refuses two ranks on one device and is skipped)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_reh
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --size 2e7 --steps 30 --windows 3 --no-cpu-baseline > $OUT/rehearse2.json 2> $OUT/rehearse2.err; echo "rehearse rc=$?"
python3 -c "
import json; d=json.loads(open('$OUT/rehearse2.json').read().strip().splitlines()[-1]); print(d['value'], d['config'].get('comm'), d.get('transports'), d.get('transports_failed'))"
grep -v "^\[W\|Gloo\|amdgpu.ids" $OUT/rehearse2.err | tail -5
timeout -k 10 400 python3 bench.py --gpus 2 --backend gloo --workload c1c --steps 30 --windows 2 --no-cpu-baseline > $OUT/rehearse2_chain.json 2> $OUT/rehearse2_chain.err; echo "rehearse chain rc=$?"
tail -c 600 $OUT/rehearse2_chain.json
