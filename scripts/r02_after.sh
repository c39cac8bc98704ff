#!/bin/bash
# "after" evidence: kernel-trace gap tables (config 2, shard size) with the round-2 build + the N = 2 rehearsals of bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_after
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --size 2e7 --steps 30 --windows 3 --no-cpu-baseline > $OUT/rehearse2.json 2> $OUT/rehearse2.err; echo "rehearse rc=$?"; python3 -c "
import json; d=json.loads(open('$OUT/rehearse2.json').read().strip().splitlines()[-1]); print(d['value'], d['config']['comm'], d['transports'])"; grep -v "^\[W\|Gloo\|amdgpu.ids" $OUT/rehearse2.err | tail -5
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --backend gloo --comm torch --size 2e7 --steps 30 --windows 2 --no-cpu-baseline > $OUT/rehearse2t.json 2> $OUT/rehearse2t.err; echo "rehearse torch rc=$?"; tail -c 300 $OUT/rehearse2t.json
export CGO_BENCH_NO_PROFILE=1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 10 --windows 1 > $OUT/trace_c2.log 2>&1; echo "trace c2 rc=$?"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_shard -- python3 $R/bench.py --size 1.25e7 --steps 300 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/trace_shard.log 2>&1; echo "trace shard rc=$?"
unset CGO_BENCH_NO_PROFILE
cd $R
python3 scripts/gap_table.py $OUT/trace_c2 --skip 60 --out $OUT/gaps_c2.json > $OUT/gaps_c2.txt 2>&1; tail -12 $OUT/gaps_c2.txt
python3 scripts/gap_table.py $OUT/trace_shard --skip 60 --out $OUT/gaps_shard.json > $OUT/gaps_shard.txt 2>&1; tail -12 $OUT/gaps_shard.txt
find $OUT -name '*kernel_trace.csv' -size +20M -delete
