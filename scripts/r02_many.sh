#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_many
mkdir -p $OUT
cd $R
timeout -k 10 200 scripts/tune/rw_mix 1e7 15 many > $OUT/many_1e7.log 2>&1; echo rc=$?; cat $OUT/many_1e7.log
