#!/bin/bash
# Round-3 profile collection on the GPU box (everything under gpurun_out/r03_prof): bench lines of every BASELINE config with
# the CPU baselines beside them, rocprofv3 kernel stats for each, separate FETCH_SIZE / WRITE_SIZE PMC passes for the headline
# config, the kernel-trace gap table of the 8-GPU shard size.
# Afterwards, in the container:  python3 scripts/summarize_profiles.py r03 r03_prof && python3 scripts/regen_tables.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_prof
mkdir -p $OUT
export TMPDIR=/tmp
# PMC passes FIRST, condensed on the box: bench.py attaches roofline.traffic only from a summary of the SAME library build
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- python3 $R/bench.py --steps 6 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- python3 $R/bench.py --steps 6 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1; echo "pmc write rc=$?"
(cd $R && python3 scripts/summarize_profiles.py r03 r03_prof > $OUT/summarize_on_box.log 2>&1; echo "summarize rc=$?")
cd $R && python3 bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench c5 rc=$?"; tail -c 300 $OUT/bench_n1.json
python3 bench.py --workload c1 --steps 15 --warmup 3 --windows 1 > $OUT/bench_c1.json 2> $OUT/bench_c1.err; echo "bench c1 rc=$?"
python3 bench.py --workload c1c --steps 15 --warmup 3 --windows 1 > $OUT/bench_c1c.json 2> $OUT/bench_c1c.err; echo "bench c1c rc=$?"
python3 bench.py --workload c2 --steps 200 --warmup 10 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 rc=$?"
CGO_RESIDENT=0 python3 bench.py --workload c2 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_c2_hostdriven.json 2> $OUT/bench_c2_hostdriven.err; echo "bench c2 host-driven rc=$?"
CGO_RESIDENT=0 python3 bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/bench_c1_hostdriven.json 2> $OUT/bench_c1_hostdriven.err; echo "bench c1 host-driven rc=$?"
python3 bench.py --workload c3 --steps 200 --warmup 10 --windows 2 > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "bench c3 rc=$?"
python3 bench.py --workload c4 --steps 45 --warmup 10 --windows 2 > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 rc=$?"
CGO_LBFGS_SPEC=0 python3 bench.py --workload c4 --steps 45 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/bench_c4_twopass.json 2> $OUT/bench_c4_twopass.err; echo "bench c4 two-pass rc=$?"
python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 5 --no-cpu-baseline > $OUT/bench_shard.json 2> $OUT/bench_shard.err; echo "bench shard rc=$?"
# N > 1 rehearsal on ONE GPU (gloo rendezvous, mailbox transport; RCCL cannot place several ranks on one device)
python3 bench.py --gpus 2 --backend gloo --size 4e7 --steps 30 --warmup 5 --windows 3 > $OUT/bench_rehearsal_2ranks.json 2> $OUT/bench_rehearsal_2ranks.err; echo "rehearsal 2 ranks rc=$?"
python3 bench.py --gpus 4 --backend gloo --size 4e7 --steps 30 --warmup 5 --windows 3 > $OUT/bench_rehearsal_4ranks.json 2> $OUT/bench_rehearsal_4ranks.err; echo "rehearsal 4 ranks rc=$?"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_stats.log 2>&1; echo "stats c5 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_c1 -- python3 $R/bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/prof_stats_c1.log 2>&1; echo "stats c1 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_c2 -- python3 $R/bench.py --workload c2 --steps 200 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/prof_stats_c2.log 2>&1; echo "stats c2 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_c3 -- python3 $R/bench.py --workload c3 --steps 200 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/prof_stats_c3.log 2>&1; echo "stats c3 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats_c4 -- python3 $R/bench.py --workload c4 --steps 45 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/prof_stats_c4.log 2>&1; echo "stats c4 rc=$?"
CGO_BENCH_NO_PROFILE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_shard -- python3 $R/bench.py --size 1.25e7 --steps 300 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/trace_shard.log 2>&1; echo "trace shard rc=$?"
(cd $R && python3 scripts/gap_table.py $OUT/trace_shard --skip 60 --out $OUT/gaps_shard.json > $OUT/gaps_shard.txt 2>&1; tail -8 $OUT/gaps_shard.txt)
# the merged-back output is capped: drop the per-dispatch traces, keep stats + counters
find $OUT -name '*kernel_trace.csv' -size +5M -delete
find $OUT -type f | wc -l
