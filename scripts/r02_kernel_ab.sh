#!/bin/bash
# Round 2, kernel A/B on one MI355X: FMA partial sums + ping-pong x/u for BIG launches + grid sweep at the shard size.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_ab
mkdir -p $OUT
cd $R
python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && grep -n "^FAILED\|^ERROR" $OUT/pytest_gpu.log | head -20
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
if 'kernels' in d:
    print('$2', 'value %.1f it/s'%d['value'], 'trials/iter %.2f launches/iter %.2f'%(d['config']['trials_per_iteration'], d['config']['launches_per_iteration']), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'])) for k,v in d['kernels'].items()})
else:
    print('$2', d)
"; }
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/c5_pp.json 2>$OUT/c5_pp.err; show $OUT/c5_pp.json "c5 pingpong"
CGO_PINGPONG=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/c5_inplace.json 2>$OUT/c5_inplace.err; show $OUT/c5_inplace.json "c5 in-place"
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/c5_pp100.json 2>$OUT/c5_pp100.err; show $OUT/c5_pp100.json "c5 pingpong 100 steps"
for g in 512 1024 2048 4096; do
  CGO_GRID_CG7=$g python3 bench.py --size 1.25e7 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/shard_g$g.json 2>$OUT/shard_g$g.err; show $OUT/shard_g$g.json "shard 1.25e7 grid7=$g"
done
for g in 512 1024 2048; do
  CGO_GRID_CG7=$g python3 bench.py --size 3e7 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/n3e7_g$g.json 2>$OUT/n3e7_g$g.err; show $OUT/n3e7_g$g.json "n=3e7 grid7=$g"
done
for w in "c2 200" "c3 200" "c4 90"; do set -- $w
  python3 bench.py --workload $1 --steps $2 --warmup 10 > $OUT/$1.json 2>$OUT/$1.err; show $OUT/$1.json "$1"
done
# Rosenbrock with more points now that the sums are fused
for p in 3 5 7; do
  m5=9000000000000000000; m7=9000000000000000000; [ $p -ge 5 ] && m5=0; [ $p -ge 7 ] && m7=0
  CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=$m5 CGO_MULTI7_MIN_N=$m7 python3 bench.py --workload c3 --steps 200 --warmup 10 > $OUT/c3_p$p.json 2>$OUT/c3_p$p.err; show $OUT/c3_p$p.json "c3 points=$p"
done
# bench.py N > 1 flow, rehearsed with 2 ranks on ONE GPU (gloo rendezvous; RCCL cannot put two ranks on one device)
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --size 2e7 --steps 30 --windows 3 --no-cpu-baseline > $OUT/rehearse2.json 2> $OUT/rehearse2.err; echo "rehearse rc=$?"; tail -c 1500 $OUT/rehearse2.json; tail -5 $OUT/rehearse2.err
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --comm torch --size 2e7 --steps 30 --windows 2 --no-cpu-baseline > $OUT/rehearse2t.json 2> $OUT/rehearse2t.err; echo "rehearse torch rc=$?"; tail -c 600 $OUT/rehearse2t.json; tail -3 $OUT/rehearse2t.err
