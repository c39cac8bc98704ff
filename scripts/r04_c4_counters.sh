#!/bin/bash
# Why config 4's launch (22 read + 5 write streams) stops at 0.67 of 8 TB/s where config 5's (3 + 2 streams) reaches 0.76: the
# same TCC / EA counters for both dominant kernels, per launch.  One --pmc group per pass, the program directly after `--`.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_c4pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
G1="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
G2="TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
G3="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum"
G4="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for g in "$G1" "$G2" "$G3" "$G4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/c4_g$i -- python3 $R/bench.py --workload c4 --steps 30 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/c4_g$i.log 2>&1; echo "c4 g$i rc=$?"
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $OUT/c5_g$i -- python3 $R/bench.py --steps 10 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/c5_g$i.log 2>&1; echo "c5 g$i rc=$?"
done
cd $R
python3 scripts/pmc_table.py $OUT/c4.json $OUT/c4_g1 $OUT/c4_g2 $OUT/c4_g3 $OUT/c4_g4 --skip 2 --match k_lbfgs_combine_spec > $OUT/c4_table.txt 2>&1; tail -3 $OUT/c4_table.txt
python3 scripts/pmc_table.py $OUT/c5.json $OUT/c5_g1 $OUT/c5_g2 $OUT/c5_g3 $OUT/c5_g4 --skip 2 --match "k_cg<" > $OUT/c5_table.txt 2>&1; tail -3 $OUT/c5_table.txt
find $OUT -name '*kernel_trace.csv' -size +2M -delete; find $OUT -name '*.db' -delete
