#!/bin/bash
# VALU-priority time slicing between the two workgroups of a CU (CGO_PRIO_SHIFT): in-kernel timelines + bench at n = 1.25e7
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_prio
mkdir -p $OUT
cd $R
export CGO_PLACE_TUNE=0
BIGN=9000000000000000000
st() { tag=$1; shift; env CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_stamps.so CGO_STAMPS_OUT=$OUT/$tag.npz "$@" python3 scripts/r04_stamps.py 12500000 40 > $OUT/$tag.txt 2>&1; echo "== $tag rc=$?"; sed -n "1,11p" $OUT/$tag.txt; grep "CUs with two" $OUT/$tag.txt; }
for s in 0 -1 5 6 7 8 10 12; do st p7_prio$s CGO_PRIO_SHIFT=$s; done
st p3_prio7 CGO_PRIO_SHIFT=7 CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN CGO_GRID_SMALL=512
st p1_prio7 CGO_PRIO_SHIFT=7 CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN CGO_GRID_SMALL=512
be() { tag=$1; shift; env "$@" python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 5 --no-cpu-baseline > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  echo "== bench $tag: $(python3 -c "import json; d=json.load(open('$OUT/b_$tag.json')); k=d['kernels']; print(round(d['value']), round(d['value_median']), 'it/s;', {n: (v['launches'], round(v['avg_us'],1)) for n,v in k.items()})")"; }
for s in 0 6 7 8 10 0 7; do be prio$s CGO_PRIO_SHIFT=$s; done
for s in 0 7; do be c3_prio$s CGO_PRIO_SHIFT=$s CGO_GRID_SMALL=512; done
