#!/bin/bash
# one box sample: bare R/W mix in place vs out of place, and the engine's dominant launch in place vs ping-pong
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_pp
mkdir -p $OUT
cd $R
T=$(date +%s)
timeout -k 10 100 scripts/tune/rw_mix 1e8 7 > $OUT/mix_$T.log 2>&1
a=$(grep "R3W2 in place      chunk/WG    U2 thr256  ntL ntS  grid= 4096" $OUT/mix_$T.log | sed 's/.*med *\([0-9.]*\) us.*/\1/')
b=$(grep "R3W2 out of place  chunk/WG    U2" $OUT/mix_$T.log | sed 's/.*med *\([0-9.]*\) us.*/\1/')
c=$(grep "copy R1W1          chunk/WG    U2" $OUT/mix_$T.log | sed 's/.*med *\([0-9.]*\) us.*/\1/')
e1=$(CGO_PINGPONG=0 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['accept_dir_trial']['avg_us'],1), round(d['value_median'],1))")
e2=$(CGO_PINGPONG=1 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['accept_dir_trial']['avg_us'],1), round(d['value_median'],1))")
e3=$(CGO_PINGPONG=0 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['accept_dir_trial']['avg_us'],1), round(d['value_median'],1))")
echo "BOX $T: mix in-place $a us, out-of-place $b us, copy $c us | engine in-place $e1 | ping-pong $e2 | in-place again $e3" | tee $OUT/sample_$T.txt
