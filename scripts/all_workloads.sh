#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for w in "c2 --steps 200" "c3 --steps 200" "c4 --steps 90" "c5 --steps 100 --no-cpu-baseline"; do
  python3 bench.py --workload $w --warmup 10 2>gpurun_out/wl.err | python3 -c "
import json,sys
txt=sys.stdin.read().strip()
if not txt: print('FAILED'); print(open('gpurun_out/wl.err').read()[-1500:]); sys.exit(0)
d=json.loads(txt.splitlines()[-1])
print(d['config']['workload'][:60]); print('   value %.1f it/s  ms/step %.3f trials/iter %.2f launches/iter %.2f  all-kernel GB/s %.0f  kernel frac %.2f  roofline %s %.0f GB/s (%.1f%%)'%(d['value'],d['ms_per_step'],d['config']['trials_per_iteration'],d['config']['launches_per_iteration'],d['achieved_hbm_gbps_per_gpu_all_kernels'],d['kernel_time_fraction_of_wall'],d['roofline']['kernel'],d['roofline']['achieved'],100*d['roofline']['frac']))
for k,v in d['kernels'].items(): print('      %-18s %6d launches avg %9.1f us %8.0f GB/s'%(k,v['launches'],v['avg_us'],v['gbps']))
"
done
