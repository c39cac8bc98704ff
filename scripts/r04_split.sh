#!/bin/bash
# uneven static split between the two workgroups of a CU (CGO_SPLIT_W7 / _W5): sweep at the 8-GPU shard size
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_split
mkdir -p $OUT
cd $R
L=$R/conjugategradientoptim.jl_amd/lib
BIGN=9000000000000000000
export CGO_PLACE_TUNE=0
st() { tag=$1; shift; env CGO_LIB_PATH=$L/libcgo_hip_stamps.so CGO_STAMPS_OUT=$OUT/$tag.npz "$@" python3 scripts/r04_stamps.py ${N:-12500000} 40 > $OUT/$tag.txt 2>&1; echo "== $tag rc=$?"; sed -n "1,2p;6,6p;10,10p" $OUT/$tag.txt; grep "CUs with two" $OUT/$tag.txt; }
be() { tag=$1; shift; env "$@" python3 bench.py --size ${N:-1.25e7} --steps 100 --warmup 10 --windows 5 --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  echo "== $tag: $(python3 -c "import json; d=json.load(open('$OUT/$tag.json')); k=d['kernels']; print(round(d['value']), round(d['value_median']), 'it/s;', {n: (v['launches'], round(v['avg_us'],1)) for n,v in k.items()})")"; }
for w in 0 0.56 0.60 0.64 0.68; do st st_w$w CGO_SPLIT_W7=$w; done
for w in 0 0.54 0.58 0.60 0.62 0.64 0.66 0.70 0 0.62; do be b_w$w CGO_SPLIT_W7=$w; done
for w in 0 0.52 0.55 0.58; do be b5_w$w CGO_MULTI7_MIN_N=$BIGN CGO_SPLIT_W5=$w; done
N=6e6; for w in 0 0.58 0.62 0.66; do N=6e6 be n6e6_w$w CGO_SPLIT_W7=$w; done
N=2.5e7; for w in 0 0.58 0.62 0.66; do N=2.5e7 be n2p5e7_w$w CGO_SPLIT_W7=$w; done
