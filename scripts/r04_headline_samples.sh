#!/bin/bash
# The headline command in 16 fresh processes: which placement level does a process get, with the (opt-in) search?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_samples
mkdir -p $OUT
cd $R
for i in $(seq 1 16); do
  timeout -k 10 120 python3 bench.py --steps 30 --warmup 5 --windows 2 --no-cpu-baseline > $OUT/s$i.json 2> $OUT/s$i.err
  python3 - <<PY
import json
d=json.loads(open("$OUT/s$i.json").read().strip().splitlines()[-1])
p=d.get("placement") or {}
print(f"process $i: {d['value']:7.1f} it/s [median {d['value_median']:7.1f}]  {d['roofline']['kernel']} {d['roofline']['avg_launch_us']:6.1f} µs = {d['roofline']['frac']:.3f} of 8 TB/s   placement: {p.get('level')} ({p.get('candidates')} candidates, mix {p.get('mix_as_allocated_us',0):.0f} -> {p.get('mix_chosen_us',0):.0f} µs)")
PY
done
