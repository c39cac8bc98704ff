#!/bin/bash
# A/B on one box: default instruction scheduling vs -amdgpu-sched-strategy=max-ilp (libcgo_hip_ilp.so) on every BASELINE workload
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_ilp
mkdir -p $OUT
cd $R
ILP=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_ilp.so
be() { tag=$1; lib=$2; shift 2; if [ "$lib" = ilp ]; then export CGO_LIB_PATH=$ILP; else unset CGO_LIB_PATH; fi
  python3 bench.py "$@" --no-cpu-baseline > $OUT/$tag.json 2> $OUT/$tag.err
  echo "== $tag: $(python3 -c "import json; d=json.load(open('$OUT/$tag.json')); k=d['kernels']; print(round(d['value']), round(d['value_median']), 'it/s;', {n: (v['launches'], round(v['avg_us'],1)) for n,v in k.items()}, d.get('placement'))")"; }
for rep in 1 2; do
  for lib in def ilp; do
    be shard_${lib}_$rep $lib --size 1.25e7 --steps 100 --warmup 10 --windows 5
  done
done
for lib in def ilp def ilp; do be c5_${lib}_$RANDOM $lib --steps 40 --warmup 5 --windows 3; done
for lib in def ilp; do CGO_PLACE_TUNE=0 be c5nt_${lib} $lib --steps 40 --warmup 5 --windows 3; done
for lib in def ilp def ilp; do be c3_${lib}_$RANDOM $lib --workload c3 --steps 200 --warmup 10 --windows 3; done
for lib in def ilp; do be c4_${lib} $lib --workload c4 --steps 45 --warmup 10 --windows 2; done
for lib in def ilp; do be c2_${lib} $lib --workload c2 --steps 200 --warmup 10 --windows 3; done
for lib in def ilp; do be c1_${lib} $lib --workload c1 --steps 15 --warmup 3 --windows 1; done
for lib in def ilp; do CGO_RESIDENT=0 be c2h_${lib} $lib --workload c2 --steps 200 --warmup 10 --windows 3; done
for n in 3e6 3e7; do for lib in def ilp; do be n${n}_${lib} $lib --size $n --steps 100 --warmup 10 --windows 3; done; done
export CGO_LIB_PATH=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_stamps.so
