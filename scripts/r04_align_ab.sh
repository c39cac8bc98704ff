#!/bin/bash
# Chunk alignment of the pure-HBM launches (1 = plain ceiling, 8 = whole 128-B lines, 16, 64 pairs): same box, alternating.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_align_ab
mkdir -p $OUT
cd $R
one() { # tag args…
  tag=$1; shift
  timeout -k 10 200 python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/$tag.json").read().strip().splitlines()[-1]); p=d.get("placement") or {}
print(f"  $tag: {d['value']:7.0f} [{d['value_median']:7.0f}] it/s  {d['roofline']['kernel']} {d['roofline']['avg_launch_us']:6.1f} us frac {d['roofline']['frac']:.3f}  placement {p.get('level')} cand {p.get('candidates')} mix {p.get('mix_as_allocated_us',0):.0f}->{p.get('mix_chosen_us',0):.0f}")
PY
}
for rep in 1 2 3; do for a in 1 8 16 64; do
  export CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_a$a.so
  one c5_a${a}_$rep --steps 30 --warmup 5 --windows 2
  one c5ns_a${a}_$rep --steps 30 --warmup 5 --windows 2 --no-placement-search
  one q7e7_a${a}_$rep --size 7e7 --steps 30 --warmup 5 --windows 2 --no-placement-search
  one q5e7_a${a}_$rep --size 5e7 --steps 30 --warmup 5 --windows 2 --no-placement-search
  one c4_a${a}_$rep --workload c4 --steps 45 --warmup 10 --windows 2
done; done
