#!/bin/bash
# Resident solver: its GPU tests, then BASELINE configs 1 / 1c / 2 with it on and off.  Output: gpurun_out/r03_res
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_res
mkdir -p $OUT
cd $R
timeout -k 10 700 python3 -m pytest tests/test_resident.py -x -q --durations=8 > $OUT/pytest_resident.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 $OUT/pytest_resident.log
[ $rc -ne 0 ] && { grep -n "^FAILED\|^ERROR\|^E  " $OUT/pytest_resident.log | head -30; exit 0; }
for w in c1 c1c c2; do
  steps=200; [ $w != c2 ] && steps=15
  win=5; [ $w != c2 ] && win=1
  for res in 1 0; do
    CGO_RESIDENT=$res timeout -k 10 300 python3 bench.py --workload $w --steps $steps --warmup 3 --windows $win --no-cpu-baseline > $OUT/bench_${w}_res${res}.json 2> $OUT/bench_${w}_res${res}.err
    echo "$w resident=$res rc=$? $(python3 -c "import json,sys; d=json.load(open('$OUT/bench_${w}_res${res}.json')); print(round(d['value']), 'it/s  median', round(d['value_median']), d['config']['launches_per_iteration'])" 2>&1 | tail -1)"
  done
done
