#!/bin/bash
# one box sample: the engine's headline run with and without the placement search
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_place_engine
mkdir -p $OUT
cd $R
T=$(date +%s)
eng() { CGO_PLACE_TUNE=$1 python3 bench.py --steps 40 --warmup 5 --windows 2 --no-cpu-baseline 2>$OUT/err_$T.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['kernels']['accept_dir_trial']['avg_us'],1), round(d['value'],1), round(d['value_median'],1), 'frac', round(d['roofline']['frac'],3), 'of mix', round(d['roofline'].get('frac_of_measured_mix',0),3), d.get('placement'))"; }
e1=$(eng 1); e0=$(eng 0); e1b=$(eng 1)
echo "BOX $T: tuned: $e1 | untuned: $e0 | tuned again: $e1b" | tee $OUT/sample_$T.txt; tail -2 $OUT/err_$T.txt
