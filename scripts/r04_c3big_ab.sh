#!/bin/bash
# c3 at n = 4e7 (1.28 GB per launch: just under the 1.4 GB pure-HBM threshold): grid-stride + default cache policy (library)
# against contiguous chunks + non-temporal accesses (CGO_BIG_BYTES=1e9), alternating on one box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_c3big
mkdir -p $OUT
cd $R
W="--workload c3 --size 4e7 --steps 100 --warmup 10 --windows 3 --no-cpu-baseline"
for rep in 1 2; do
  for v in lib big bigns; do
    if [ $v = lib ]; then E=""; X=""; elif [ $v = big ]; then E="CGO_BIG_BYTES=1000000000"; X=""; else E="CGO_BIG_BYTES=1000000000"; X="--no-placement-search"; fi
    env $E timeout -k 10 200 python3 bench.py $W $X > $OUT/${v}_$rep.json 2> $OUT/${v}_$rep.err; echo "$v rep=$rep rc=$?"
    python3 - <<PY
import json
d=json.loads(open("$OUT/${v}_$rep.json").read().strip().splitlines()[-1])
print("  it/s", round(d["value"]), "median", round(d.get("value_median") or 0), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],1), "frac", round(d["roofline"]["frac"],3), d.get("placement"))
PY
  done
done
