#!/bin/bash
# GPU test tier; output under gpurun_out/r03_tests.  Usage: scripts/r03_gpu_tests.sh [pytest args…]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_tests
mkdir -p $OUT
cd $R
python3 -m pytest tests -m gpu -q --durations=15 "$@" > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && grep -n "^FAILED\|^ERROR\|^E  " $OUT/pytest_gpu.log | head -40
exit 0
