#!/bin/bash
# One headline run + what rocm-smi says about the box (clocks, memory, power cap): does the "slow" placement level go with a box property?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_probe
mkdir -p $OUT
cd $R
T=$(date +%s%N | cut -c10-16)
(rocm-smi --showclocks --showmemuse --showmeminfo vram --showpower --showperflevel --showmaxpower 2>&1 | head -60) > $OUT/smi_before_$T.txt
python3 bench.py --no-cpu-baseline > $OUT/n1_$T.json 2> $OUT/n1_$T.err; echo "rc=$?"
(rocm-smi --showclocks --showpower 2>&1 | head -40) > $OUT/smi_after_$T.txt
python3 -c "
import json; d=json.load(open('$OUT/n1_$T.json')); print(round(d['value']), round(d['roofline']['frac'],3), d['placement'])"
grep -i "mclk\|sclk\|fclk\|socclk" $OUT/smi_before_$T.txt | head -8
grep -i "mclk\|power" $OUT/smi_after_$T.txt | head -6
