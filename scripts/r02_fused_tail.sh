#!/bin/bash
# Fused reduction tail (finish_tail) against the finalize launches it replaces: parity tests that cross the two paths, then
# A/B of the BASELINE configs with CGO_FUSED_TAIL=0/1 on the same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_ft
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu2.log 2>&1; rc=$?; echo "pytest (second pass) rc=$rc"; tail -2 $OUT/pytest_gpu2.log
[ $rc -ne 0 ] && exit 1
CGO_FUSED_TAIL=0 timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "ctl or controller or parity or chained" > $OUT/pytest_gpu_unfused.log 2>&1; rc=$?; echo "pytest unfused rc=$rc"; tail -3 $OUT/pytest_gpu_unfused.log
[ $rc -ne 0 ] && exit 1
run() {  # tag, env, args
    local tag=$1 ft=$2; shift 2
    CGO_FUSED_TAIL=$ft timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag failed"; tail -3 $OUT/$tag.err; return 1; }
    python3 - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = {n: (v["launches"], round(v["avg_us"], 1)) for n, v in d.get("kernels", {}).items()}
print(sys.argv[2], "value %.1f med %.1f it/s" % (d["value"], d.get("value_median") or 0), k, d.get("placement"))
PY
}
for ft in 0 1 0 1; do
    run c2_ft$ft $ft --workload c2 --steps 300 --warmup 10 --windows 3 || exit 1
done
for ft in 0 1; do
    run c3_ft$ft $ft --workload c3 --steps 100 --warmup 5 --windows 2 || exit 1
    run shard_ft$ft $ft --size 1.25e7 --steps 300 --warmup 10 --windows 3 || exit 1
    run n1e4_ft$ft $ft --size 1e4 --steps 300 --warmup 10 --windows 3 || exit 1
    run c5_ft$ft $ft --steps 20 --warmup 5 --windows 3 || exit 1
done
export CGO_BENCH_NO_PROFILE=1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_c2 -- python3 $R/bench.py --workload c2 --steps 300 --warmup 10 --windows 1 --no-cpu-baseline > $OUT/trace_c2.log 2>&1; echo "trace c2 rc=$?"
cd $R
python3 scripts/gap_table.py $OUT/trace_c2 --skip 60 --out $OUT/gaps_c2.json > $OUT/gaps_c2.txt 2>&1; tail -12 $OUT/gaps_c2.txt
find $OUT -name '*kernel_trace.csv' -size +20M -delete
echo done
