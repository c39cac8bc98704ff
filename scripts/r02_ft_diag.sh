#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02_ft
mkdir -p $OUT
cd $R
CGO_TAIL_STRICT=1 timeout -k 10 600 python3 -m pytest tests -q -m gpu -x > $OUT/pytest_strict.log 2>&1; echo "strict rc=$?"; tail -3 $OUT/pytest_strict.log
timeout -k 10 600 python3 -m pytest tests -q -m gpu > $OUT/pytest_light.log 2>&1; echo "light rc=$?"; grep -E "^FAILED|passed|failed" $OUT/pytest_light.log | head -40
for s in 1 0; do
CGO_TAIL_STRICT=$s timeout -k 10 300 python3 bench.py --workload c2 --steps 300 --warmup 10 --windows 3 --no-cpu-baseline > $OUT/c2_strict$s.json 2> $OUT/c2_strict$s.err; python3 -c "
import json; d=json.loads(open('$OUT/c2_strict$s.json').read().strip().splitlines()[-1]); print('strict=$s', d['value'], d.get('value_median'), {n:(v['launches'], round(v['avg_us'],1)) for n,v in d['kernels'].items()})"
done
