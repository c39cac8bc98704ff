#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_arena
mkdir -p $OUT
cd $R
T=$(date +%s)
timeout -k 10 200 scripts/tune/rw_mix 1e8 7 arena > $OUT/arena_$T.log 2>&1; echo rc=$?; cat $OUT/arena_$T.log | sed 's/n=1.00e+08 R3W2 in place      chunk\/WG    U2 thr256  ntL ntS  grid= 4096 //'
