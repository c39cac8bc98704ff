#!/bin/bash
# dynamic instruction mix of a resident slice (config 1): where do 14.7 us per iteration go?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_respmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -- python3 $R/bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name '*counter_collection.csv' | head -1)
  echo "== $set"; [ -n "$f" ] && grep k_resident $f | awk -F',' '{print $(NF-1), $NF}' | tr -d '"' | sort | uniq -c | head -12
done
