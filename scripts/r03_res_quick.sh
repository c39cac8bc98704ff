#!/bin/bash
# quick look: configs 1 / 2 (and n = 2e5) on the resident solver, no instrumentation, events off
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_resq
mkdir -p $OUT
cd $R
run() { tag=$1; shift; CGO_BENCH_NO_PROFILE=1 "$@" > $OUT/$tag.json 2> $OUT/$tag.err; echo "== $tag: $(python3 -c "import json; d=json.load(open('$OUT/$tag.json')); print(round(d['value']), round(d['value_median']), round(d['value_max']))")"; }
run c1 python3 bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline
run c1_p7 env CGO_RES_POINTS=7 python3 bench.py --workload c1 --steps 15 --warmup 3 --windows 1 --no-cpu-baseline
run c1_dy python3 bench.py --workload c1 --beta DaiYuan --steps 200 --warmup 3 --windows 3 --no-cpu-baseline
run c2 python3 bench.py --workload c2 --steps 200 --warmup 3 --windows 5 --no-cpu-baseline
run c2_p1 env CGO_RES_POINTS=1 python3 bench.py --workload c2 --steps 200 --warmup 3 --windows 5 --no-cpu-baseline
run c2_2e5 python3 bench.py --workload c2 --size 2e5 --steps 200 --warmup 3 --windows 5 --no-cpu-baseline
run c2_2e4 python3 bench.py --workload c2 --size 2e4 --steps 200 --warmup 3 --windows 5 --no-cpu-baseline
run c2_4e3 python3 bench.py --workload c2 --size 4e3 --steps 200 --warmup 3 --windows 5 --no-cpu-baseline
