#!/bin/bash
# config 4 (LSE n = 1e7, L-BFGS m = 10): first trial fused into the direction pass on/off
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_c4
mkdir -p $OUT
cd $R
for f in 1 0 1 0; do
  CGO_LBFGS_FUSE_TRIAL=$f python3 bench.py --workload c4 --steps 45 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c4_f$f.json 2> $OUT/c4_f$f.err
  echo "== fuse=$f: $(python3 -c "import json; d=json.load(open('$OUT/c4_f$f.json')); print(round(d['value']), round(d['value_median']), 'it/s; launches/it', round(d['config']['launches_per_iteration'],2), {k: (v['launches'], round(v['avg_us'],1)) for k,v in d['kernels'].items()})")"
done
