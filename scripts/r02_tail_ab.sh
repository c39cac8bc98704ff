#!/bin/bash
# same-box A/B: transpose-reduce tail (default lib) vs round-1 tree tail (libcgo_hip_treetail.so), alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_tail_ab
mkdir -p $OUT
cd $R
show() { python3 -c "
import json,sys
try:
    d=json.loads(open('$1').read().strip().splitlines()[-1])
except Exception as e:
    print('$2 FAILED', open('$1'.replace('.json','.err')).read()[-400:]); sys.exit(0)
print('$2', 'value %.1f med %.1f it/s'%(d['value'],d['value_median']), {k:(v['launches'],round(v['avg_us'],1),round(v['gbps'] or 0)) for k,v in d['kernels'].items()})
"; }
TT=$R/conjugategradientoptim.jl_amd/lib/libcgo_hip_treetail.so
for rep in 1 2; do
  python3 bench.py --steps 50 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/c5_new_$rep.json 2>$OUT/c5_new_$rep.err; show $OUT/c5_new_$rep.json "c5 new tail  #$rep"
  CGO_LIB_PATH=$TT python3 bench.py --steps 50 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/c5_old_$rep.json 2>$OUT/c5_old_$rep.err; show $OUT/c5_old_$rep.json "c5 tree tail #$rep"
done
for n in 3e7 1.25e7; do
  python3 bench.py --size $n --steps 100 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/n${n}_new.json 2>$OUT/n${n}_new.err; show $OUT/n${n}_new.json "n=$n new tail"
  CGO_LIB_PATH=$TT python3 bench.py --size $n --steps 100 --warmup 5 --windows 3 --no-cpu-baseline > $OUT/n${n}_old.json 2>$OUT/n${n}_old.err; show $OUT/n${n}_old.json "n=$n tree tail"
done
timeout -k 10 100 scripts/tune/rw_mix 1e8 7 > $OUT/rw_mix.log 2>&1; grep -n "in place.*chunk/WG    U2 thr256  ntL ntS  grid= 4096\|out of place.*U2" $OUT/rw_mix.log
