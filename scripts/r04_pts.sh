#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_pts
mkdir -p $OUT
cd $R
export CGO_PLACE_TUNE=0
BIGN=9000000000000000000
st() { tag=$1; shift; env CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_stamps.so CGO_STAMPS_OUT=$OUT/$tag.npz "$@" python3 scripts/r04_stamps.py 12500000 40 > $OUT/$tag.txt 2>&1; echo "== $tag rc=$?"; sed -n "1,2p;6,6p;10,10p" $OUT/$tag.txt; grep "CUs with two" $OUT/$tag.txt; }
st p1 CGO_MULTI_MIN_N=$BIGN CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN CGO_GRID_SMALL=512
st p3 CGO_MULTI5_MIN_N=$BIGN CGO_MULTI7_MIN_N=$BIGN CGO_GRID_SMALL=512
st p5 CGO_MULTI7_MIN_N=$BIGN
st p7
st rosen_p3 CGO_STAMPS_ROSEN=1 CGO_GRID_SMALL=512
