#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
CGO_RES_TIMING=1 CGO_BENCH_NO_PROFILE=1 python3 bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline 2>&1 | grep -E "cgo resident|value" | cut -c1-260 | tail -8
python3 - <<'PY'
import time, numpy as np
import cgo_amd as cgo
n = 1000
obj = cgo.RosenbrockPaired(n)
cfg = cgo.setupCGConfig(1e-200, cgo.DaiYuan(), cgo.EnableTrace(), max_iters=100000)
s = cgo.Solver(obj, cfg, cgo.setupStrongWolfeBisection(1e-5, 0.8))
s.set_x0_fill("alternate", -1.2, 1.0); s.start(); s.iterate(5)
for k in (1, 5, 15, 50, 200, 1000):
    t = time.perf_counter(); s.iterate(k); dt = time.perf_counter() - t
    print(f"iterate({k}): {dt*1e6:.1f} us = {dt*1e6/k:.2f} us per iteration, stats {s.resident_stats()}")
PY
