#!/bin/bash
# GPU test tier + (optional) extra commands; output under gpurun_out/r02_tests
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_tests
mkdir -p $OUT
cd $R
python3 -m pytest tests -m gpu -q "$@" > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && grep -n "^FAILED\|^ERROR\|^E  " $OUT/pytest_gpu.log | head -40
exit 0
