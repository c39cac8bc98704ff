#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in "c2 --size 10000 --steps 600" "c2 --size 100000 --steps 600" "c2 --steps 600" "c2 --size 3000000 --steps 400" "c2 --size 10000000 --steps 300" "c5 --steps 100" "c3 --size 1000000 --steps 600"; do
  for depth in 0 4; do
    echo -n "noprof $w depth=$depth: "; CGO_BENCH_NO_PROFILE=1 CGO_CTL_DEPTH=$depth python3 bench.py --workload $w --warmup 20 --no-cpu-baseline 2>/dev/null
  done
done
