#!/bin/bash
# After taking the 56-word record copy off the controller's single lane: armed rounds vs host-driven launches again.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_cfast
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "controller or fused" > $OUT/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $OUT/pytest.log
[ $rc -ne 0 ] && exit 1
one() {  # tag depth args...
    local tag=$1 d=$2; shift 2
    CGO_CTL_DEPTH=$d timeout -k 10 300 python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err || { echo "$tag failed"; tail -3 $OUT/$tag.err; return 1; }
    python3 -c "
import json; d=json.loads(open('$OUT/$tag.json').read().strip().splitlines()[-1]); k={n:(v['launches'], round(v['avg_us'],1)) for n,v in d.get('kernels',{}).items()}; print('$tag depth=$d value %.0f med %.0f it/s' % (d['value'], d['value_median']), k)"
}
for d in 0 8 0 8; do one c2_d$d $d --workload c2 --steps 300 --warmup 10 --windows 3 || exit 1; done
export CGO_BENCH_NO_PROFILE=1
for n in 1e4 1e5 1e6 3e6; do
  for d in 0 8; do
    one q_${n}_d$d $d --size $n --steps 300 --warmup 10 --windows 3 || exit 1
    one r_${n}_d$d $d --workload c3 --size $n --steps 200 --warmup 10 --windows 3 || exit 1
  done
done
one shard_d0 0 --size 1.25e7 --steps 300 --warmup 10 --windows 3
one shard_d8 8 --size 1.25e7 --steps 300 --warmup 10 --windows 3
echo done
