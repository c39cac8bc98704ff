#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_place
mkdir -p $OUT
cd $R
T=$(date +%s)
timeout -k 10 300 scripts/tune/rw_mix 1e8 3 place all > $OUT/place_all_$T.log 2>&1; echo rc=$?; grep -v "^T " $OUT/place_all_$T.log | sed 's/n=1.00e+08 R3W2 in place      chunk\/WG    U2 thr256  ntL ntS  grid= 4096 //' | head -40
