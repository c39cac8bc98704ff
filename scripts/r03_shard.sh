#!/bin/bash
# The 8-GPU shard of config 5 on one GPU (n = 1.25e7): placement search on/off, events on/off.  Output gpurun_out/r03_shard
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_shard
mkdir -p $OUT
cd $R
for tune in 1 0 1 0; do
  CGO_DEBUG_PLACE=1 CGO_PLACE_TUNE=$tune python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 5 --no-cpu-baseline > $OUT/shard_t$tune.json 2> $OUT/shard_t$tune.err
  echo "== tune=$tune: $(python3 -c "import json; d=json.load(open('$OUT/shard_t$tune.json')); k=d['kernels']['accept_dir_trial']; print(round(d['value']), round(d['value_median']), 'it/s; accept_dir_trial', round(k['avg_us'],1), 'us', d['placement'])")"; grep "cgo place" $OUT/shard_t$tune.err | tail -2
done
CGO_PLACE_TUNE=1 CGO_BENCH_NO_PROFILE=1 python3 bench.py --size 1.25e7 --steps 100 --warmup 10 --windows 5 --no-cpu-baseline > $OUT/shard_np.json 2>/dev/null; echo "== events off, tuned: $(cat $OUT/shard_np.json | cut -c1-150)"
for w in c3; do
  for tune in 1 0; do
  CGO_DEBUG_PLACE=1 CGO_PLACE_TUNE=$tune python3 bench.py --workload $w --steps 200 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/${w}_t$tune.json 2> $OUT/${w}_t$tune.err
  echo "== $w tune=$tune: $(python3 -c "import json; d=json.load(open('$OUT/${w}_t$tune.json')); k=d['kernels']['accept_dir_trial']; print(round(d['value']), round(d['value_median']), 'it/s; accept_dir_trial', round(k['avg_us'],1), 'us', d['placement'])")"
  done
done
