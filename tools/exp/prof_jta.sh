#!/bin/bash
# per-kernel time of the JTA train step: bash tools/exp/prof_jta.sh <out.txt> [steps]   (run on the GPU box)
out=${1:-gpurun_out/r02/jta_kernels.txt}; steps=${2:-4}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_jta
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_jta -- python $R/tools/exp/jta_step.py $steps > /tmp/prof_jta.log 2>&1
cd $R
python - "$out" "$steps" <<'PY'
import csv, glob, sys, re
out, steps = sys.argv[1], int(sys.argv[2]) + 2
f = glob.glob('/tmp/prof_jta/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = 0.0
with open(out, 'w') as o:
    o.write(open('/tmp/prof_jta.log').read()[-300:] + "\n")
    for r in rows[:40]:
        ms = float(r['TotalDurationNs']) / 1e6 / steps
        tot += ms
        name = re.sub(r'\(.*', '', r['Name'])[:110]
        o.write(f"{ms:8.2f} ms/step  {int(r['Calls'])//steps:5d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {name}\n")
    o.write(f"total (top 40) {tot:.1f} ms/step\n")
PY
