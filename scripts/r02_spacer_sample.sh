#!/bin/bash
# one box sample: is the box bimodal (rw_mix place)?  and the engine's dominant launch with x/u 0 vs 4 vs 8 GiB apart
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_spacer
mkdir -p $OUT
cd $R
T=$(date +%s)
timeout -k 10 100 scripts/tune/rw_mix 1e8 5 place > $OUT/place_$T.log 2>&1
m=$(grep "triples:" $OUT/place_$T.log)
eng() { CGO_XU_SPACER_GIB=$1 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['accept_dir_trial']['avg_us'],1), round(d['value_median'],1), round(d['roofline'].get('frac_of_measured_mix',0),3))"; }
e0=$(eng 0); e4=$(eng 4); e8=$(eng 8); e0b=$(eng 0); e4b=$(eng 4)
echo "BOX $T: $m | engine spacer 0: $e0 | 4: $e4 | 8: $e8 | 0 again: $e0b | 4 again: $e4b" | tee $OUT/sample_$T.txt
