#!/bin/bash
# Round-2 "before" evidence on one MI355X: GPU test tier, the R/W-mix ceiling harness, kernel-trace gap
# tables for config 2 (n = 1e6) and the 8-GPU shard size (n = 1.25e7), and the default bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_before
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 scripts/tune/rw_mix 1e8 11 $OUT/rw_mix_1e8.csv > $OUT/rw_mix_1e8.log 2>&1; echo "rw_mix rc=$?"
timeout -k 10 200 scripts/tune/rw_mix 1.25e7 21 $OUT/rw_mix_1p25e7.csv > $OUT/rw_mix_1p25e7.log 2>&1; echo "rw_mix small rc=$?"
export CGO_BENCH_NO_PROFILE=1
for w in "c2 200" "c3 200" "c4 90"; do set -- $w
  python3 bench.py --workload $1 --steps $2 --warmup 10 > $OUT/noprof_$1.json 2>$OUT/noprof_$1.err; echo "$1 rc=$? $(cat $OUT/noprof_$1.json)"
done
python3 bench.py --size 1.25e7 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/noprof_shard.json 2>$OUT/noprof_shard.err; echo "shard rc=$? $(cat $OUT/noprof_shard.json)"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 10 > $OUT/trace_c2.log 2>&1; echo "trace c2 rc=$?"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_shard -- python3 $R/bench.py --size 1.25e7 --steps 300 --warmup 10 --no-cpu-baseline > $OUT/trace_shard.log 2>&1; echo "trace shard rc=$?"
unset CGO_BENCH_NO_PROFILE
cd $R
python3 scripts/gap_table.py $OUT/trace_c2 --skip 60 --out $OUT/gaps_c2.json > $OUT/gaps_c2.txt 2>&1; tail -25 $OUT/gaps_c2.txt
python3 scripts/gap_table.py $OUT/trace_shard --skip 60 --out $OUT/gaps_shard.json > $OUT/gaps_shard.txt 2>&1; tail -25 $OUT/gaps_shard.txt
# keep the merged-back output small: the raw traces are big
find $OUT -name '*kernel_trace.csv' -size +20M -delete
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; head -c 700 $OUT/bench_default.json
