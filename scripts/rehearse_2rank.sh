#!/bin/bash
# 2 ranks on ONE GPU (gloo rendezvous, scalar exchange through the callback ABI): rehearses bench.py's N>1 flow.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py --steps 30 --no-cpu-baseline --size 2e7 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=1 n=2e7', round(d['value'],1), d['config']['trials_per_iteration'], {k:(v['launches'],round(v['avg_us'],1)) for k,v in d['kernels'].items()})"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 30 --size 2e7 --backend gloo --comm shm 2>gpurun_out/rehearse.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2 (one GPU, gloo) n=2e7', round(d['value'],1), d['config'], {k:(v['launches'],round(v['avg_us'],1)) for k,v in d['kernels'].items()})"
tail -5 gpurun_out/rehearse.err

timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 30 --size 2e7 --backend gloo --comm torch 2>>gpurun_out/rehearse.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2 (one GPU, gloo, torch callback) n=2e7', round(d['value'],1), d['config']['comm'])"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 60 --workload c4 --size 2e6 --backend gloo --comm shm 2>>gpurun_out/rehearse.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=2 c4 LSE/L-BFGS shm', round(d['value'],1), d['config']['comm'][:30])"
