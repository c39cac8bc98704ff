#!/bin/bash
# Controller-armed rounds as ONE launch (tail_ctl): tests, then depth / point-count policy sweep on the latency-bound configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_cf
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
run() {  # tag, env assignments..., -- args
    local tag=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag failed"; tail -3 $OUT/$tag.err; return 1; }
    python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: (v["launches"], round(v["avg_us"], 1)) for n, v in d.get("kernels", {}).items()}
print(sys.argv[2], "value %.1f med %.1f it/s armed/iter %s trials/iter %.2f" % (d["value"], d.get("value_median") or 0, d.get("controller_armed_launches_per_iteration"), d.get("trials_per_iteration") or 0), k)
PY
}
for dpt in 0 4 8 16; do
    run c2_d$dpt CGO_CTL_DEPTH=$dpt -- --workload c2 --steps 300 --warmup 10 --windows 3 || exit 1
done
run c2_d8_unfused CGO_CTL_DEPTH=8 CGO_CTL_FUSED=0 -- --workload c2 --steps 300 --warmup 10 --windows 3 || exit 1
for dpt in 0 4 8; do
    run c3_p3_d$dpt CGO_CTL_DEPTH=$dpt -- --workload c3 --steps 100 --warmup 5 --windows 2 || exit 1
    run c3_p7_d$dpt CGO_CTL_DEPTH=$dpt CGO_MULTI_MIN_N=0 CGO_MULTI5_MIN_N=0 CGO_MULTI7_MIN_N=0 -- --workload c3 --steps 100 --warmup 5 --windows 2 || exit 1
    run shard_d$dpt CGO_CTL_DEPTH=$dpt -- --size 1.25e7 --steps 300 --warmup 10 --windows 3 || exit 1
    run n1e4_d$dpt CGO_CTL_DEPTH=$dpt -- --size 1e4 --steps 300 --warmup 10 --windows 3 || exit 1
    run n1e5_d$dpt CGO_CTL_DEPTH=$dpt -- --size 1e5 --steps 300 --warmup 10 --windows 3 || exit 1
done
echo done
