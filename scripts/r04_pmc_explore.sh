#!/bin/bash
# Round 4, first GPU call: counters.
#  (1) the placement levels at n = 1e8: TCC / EA counters of ONE kernel on a fast and on a slow (x, u, D) triple of the same process
#      (scripts/tune/place_pmc.hip)                                                           → gpurun_out/r04_pmc/place_*
#  (2) the 8-GPU shard size n = 1.25e7: SQ / TCC counters of the 7-point launch, the 1-point launch and the bare stream mix
#      on the same grid                                                                        → gpurun_out/r04_pmc/shard_*
#  (3) FETCH_SIZE / WRITE_SIZE of config 3 (n = 1e7 and 4e7) and config 4                      → gpurun_out/r04_pmc/c3_*, c4_*
# One --pmc group per pass, the program directly after `--`.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/rocprofv3_list_avail.txt 2>&1; echo "list rc=$? ($(wc -l < $OUT/rocprofv3_list_avail.txt) lines)"
$R/scripts/tune/place_pmc > $OUT/place_plain.log 2>&1; echo "place plain rc=$?"; tail -3 $OUT/place_plain.log

pmc() {  # pmc TAG "COUNTERS" -- program args…   (env already exported by the caller)
  tag=$1; set=$2; shift 3
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- "$@" > $OUT/$tag.log 2>&1
  echo "pmc $tag [$set] rc=$?"
}

G_TCC1="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
G_TCC2="TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"
G_TCC3="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum"
G_TCC4="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
G_TCC5="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum"
G_TCC6="TCC_REQ_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum TCC_CYCLE_sum"
G_TCC7="TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_BUBBLE_sum"
G_SQ1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
G_SQ2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_SCA"
G_SQ3="SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VALU_INT64"

STAGES=${1:-"1 2 3"}
# ---- (1) placement levels
if [[ " $STAGES " == *" 1 "* ]]; then
i=0
for g in "$G_TCC1" "$G_TCC2" "$G_TCC3" "$G_TCC4" "$G_TCC5" "$G_TCC6" "$G_TCC7" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_WRREQ" "$G_SQ1"; do
  i=$((i+1)); pmc place_g$i "$g" -- $R/scripts/tune/place_pmc 1e8 10 6
  grep -E "FAST triple|round" $OUT/place_g$i.log | tail -3
done
(cd $R && python3 scripts/pmc_table.py $OUT/place_table.json $OUT/place_g* --match "k_mix<" > $OUT/place_table.txt 2>&1; echo "place table rc=$?")

fi
if [[ " $STAGES " == *" 2 "* ]]; then
# ---- (2) shard size: (A) default 7-point fused launch with the stream mix of the placement search beside it; (B) 1-point launch, sums by finalize launches
BIGN=9000000000000000000
i=0
for g in "$G_SQ1" "$G_SQ2" "$G_SQ3" "$G_TCC1" "$G_TCC2" "$G_TCC4" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  export CGO_PLACE_TUNE=1 CGO_PLACE_MIN_BYTES=2.7e8
  pmc shardA_g$i "$g" -- python3 $R/bench.py --size 1.25e7 --steps 60 --warmup 10 --windows 1 --no-cpu-baseline
  unset CGO_PLACE_MIN_BYTES; export CGO_PLACE_TUNE=0
  export CGO_FUSED_TAIL=0 CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN
  pmc shardB_g$i "$g" -- python3 $R/bench.py --size 1.25e7 --steps 60 --warmup 10 --windows 1 --no-cpu-baseline
  unset CGO_FUSED_TAIL CGO_MULTI_MIN_N CGO_MULTI5_MIN_N CGO_MULTI7_MIN_N CGO_PLACE_TUNE
done
(cd $R && python3 scripts/pmc_table.py $OUT/shard_table.json $OUT/shardA_g* $OUT/shardB_g* --skip 4 > $OUT/shard_table.txt 2>&1; echo "shard table rc=$?")

fi
if [[ " $STAGES " == *" 3 "* ]]; then
# ---- (3) HBM traffic of configs 3 and 4
for w in "c3 1e7 200" "c3 4e7 100" "c4 1e7 45"; do
  set -- $w
  pmc $1_n$2_fetch "FETCH_SIZE" -- python3 $R/bench.py --workload $1 --size $2 --steps $3 --warmup 10 --windows 1 --no-cpu-baseline
  pmc $1_n$2_write "WRITE_SIZE" -- python3 $R/bench.py --workload $1 --size $2 --steps $3 --warmup 10 --windows 1 --no-cpu-baseline
  (cd $R && python3 scripts/pmc_table.py $OUT/$1_n$2_table.json $OUT/$1_n$2_fetch $OUT/$1_n$2_write --skip 2 > $OUT/$1_n$2_table.txt 2>&1)
done
# un-profiled reference timings of the same commands (kernel durations by HIP events)
cd $R
python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/bench_shard.json 2> $OUT/bench_shard.err; echo "bench shard rc=$?"
python3 bench.py --workload c3 --size 4e7 --steps 100 --warmup 10 --windows 2 --no-cpu-baseline > $OUT/bench_c3_4e7.json 2> $OUT/bench_c3_4e7.err; echo "bench c3 4e7 rc=$?"
fi
find $OUT -name '*kernel_trace.csv' -size +3M -delete
find $OUT -name '*.db' -delete
du -sh $OUT
