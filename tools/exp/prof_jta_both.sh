#!/bin/bash
# per-kernel times of tools/exp/jta_time.py (fp32-class AND bf16 steps in one profile): bash tools/exp/prof_jta_both.sh <label>
mkdir -p gpurun_out/r04
R=$PWD
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pj
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pj -- python $R/tools/exp/jta_time.py > /tmp/pj.log 2>&1
F=$(find /tmp/pj -name '*kernel_stats.csv' | head -1)
python - "$F" > $R/gpurun_out/r04/jta_kernels_${1:-x}.txt <<PY
import csv,sys
for i,r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<60: print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:10.1f} us {r["Percentage"]:>6s} %')
PY
cat $R/gpurun_out/r04/jta_kernels_${1:-x}.txt; tail -5 /tmp/pj.log
