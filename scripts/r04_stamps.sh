#!/bin/bash
# In-kernel timelines (diagnostic build libcgo_hip_stamps.so) of the accept+dir+trial launch at the 8-GPU shard size.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_stamps
mkdir -p $OUT
cd $R
export CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_stamps.so CGO_PLACE_TUNE=0
BIGN=9000000000000000000
run() { tag=$1; shift; env CGO_STAMPS_OUT=$OUT/$tag.npz "$@" python3 scripts/r04_stamps.py 12500000 40 > $OUT/$tag.txt 2>&1; echo "== $tag rc=$?"; sed -n "1,11p;28,40p" $OUT/$tag.txt; }
run p7_g512
run p7_g1024 CGO_GRID_CG7=1024
run p7_g256 CGO_GRID_CG7=256
run p7_g512_unfused CGO_FUSED_TAIL=0
run p1_g512 CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN CGO_GRID_SMALL=512
run p1_g256 CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN
run p3_g512 CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN CGO_GRID_SMALL=512
