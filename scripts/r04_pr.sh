#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_pr
mkdir -p $OUT
cd $R
L=$R/conjugategradientoptim.jl_amd/lib
st() { tag=$1; lib=$2; shift 2; env CGO_PLACE_TUNE=0 CGO_LIB_PATH=$L/$lib CGO_STAMPS_OUT=$OUT/$tag.npz "$@" python3 scripts/r04_stamps.py 12500000 40 > $OUT/$tag.txt 2>&1; echo "== $tag rc=$?"; sed -n "1,2p;6,6p;10,10p" $OUT/$tag.txt; grep "CUs with two" $OUT/$tag.txt; }
st base libcgo_hip_stamps.so
st pr libcgo_hip_pr.so
st base2 libcgo_hip_stamps.so
st pr2 libcgo_hip_pr.so
be() { tag=$1; lib=$2; shift 2; CGO_LIB_PATH=$L/$lib python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  echo "== $tag: $(python3 -c "import json; d=json.load(open('$OUT/$tag.json')); k=d['kernels']; print(round(d['value']), round(d['value_median']), 'it/s;', {n: (v['launches'], round(v['avg_us'],1)) for n,v in k.items()}, d.get('placement'))")"; }
for rep in 1 2; do for lib in libcgo_hip.so libcgo_hip_pr.so; do CGO_PLACE_TUNE=0 be shard_${lib}_$rep $lib --size 1.25e7 --steps 100 --warmup 10 --windows 5; done; done
for rep in 1 2; do for lib in libcgo_hip.so libcgo_hip_pr.so; do be c5_${lib}_$rep $lib --steps 40 --warmup 5 --windows 3; done; done
for n in 1e6 3e6 3e7; do for lib in libcgo_hip.so libcgo_hip_pr.so; do CGO_RESIDENT=0 be n${n}_${lib} $lib --size $n --steps 100 --warmup 10 --windows 3; done; done
