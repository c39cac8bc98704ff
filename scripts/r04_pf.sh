#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_pf
mkdir -p $OUT
cd $R
export CGO_PLACE_TUNE=0
st() { tag=$1; lib=$2; shift 2; env CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/$lib CGO_STAMPS_OUT=$OUT/$tag.npz "$@" python3 scripts/r04_stamps.py 12500000 40 > $OUT/$tag.txt 2>&1; echo "== $tag rc=$?"; sed -n "1,11p" $OUT/$tag.txt; grep "CUs with two" $OUT/$tag.txt; }
st base libcgo_hip_stamps.so
st pf libcgo_hip_pf.so
st base2 libcgo_hip_stamps.so
st pf2 libcgo_hip_pf.so
st pf_g768 libcgo_hip_pf.so CGO_GRID_CG7=768
st base_g768 libcgo_hip_stamps.so CGO_GRID_CG7=768
