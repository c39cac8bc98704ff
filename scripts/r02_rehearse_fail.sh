#!/bin/bash
# bench.py N > 1 failure paths, 2 ranks on ONE GPU: RCCL cannot place two ranks on one device — the run must still end
# with the mailbox result, either because the communicator reports an error (skip) or because it never returns (watchdog).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_rehearse
mkdir -p $OUT
cd $R
CGO_BENCH_TRY_RCCL=1 timeout -k 10 200 python3 bench.py --gpus 2 --backend gloo --size 2e7 --steps 30 --windows 2 --transport-timeout 40 --no-cpu-baseline > $OUT/fail.json 2> $OUT/fail.err; echo "rc=$?"
python3 -c "
import json; d=json.loads(open('$OUT/fail.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['comm'], list(d['transports']), d.get('transports_failed'), d.get('note'))"
grep -v "^\[W\|Gloo\|amdgpu.ids" $OUT/fail.err | tail -12
for w in c1 c1c; do python3 bench.py --workload $w --steps 100 --warmup 5 --windows 1 > $OUT/$w.json 2>$OUT/$w.err; python3 -c "
import json; d=json.loads(open('$OUT/$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d.get('stopped_early'), d['config']['trials_per_iteration'], d['kernels'])"; tail -2 $OUT/$w.err; done
