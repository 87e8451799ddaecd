#!/bin/bash
# per-kernel time of the rollout step (env leg only): bash tools/exp/prof_env.sh <out.txt>   (run on the GPU box)
out=${1:-gpurun_out/r02/env_kernels.txt}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_env
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -- python $R/bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_policy --no_jta > /tmp/prof_env.log 2>&1
cd $R
python - "$out" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
f = glob.glob('/tmp/prof_env/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open(out, 'w') as o:
    o.write(open('/tmp/prof_env.log').read()[-1500:] + "\n")
    for r in rows[:30]:
        name = re.sub(r'\(.*', '', r['Name'])[:100]
        o.write(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms total  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {name}\n")
PY
python - "$out" <<'PY'
import csv, glob, sys, re
f = glob.glob('/tmp/prof_env/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sim_step_kernel' in r['Kernel_Name']]
a, b = idx[-100], idx[-98]
with open(sys.argv[1], 'a') as o:
    o.write("\none step, in launch order (start offset us, duration us, kernel):\n")
    t0 = int(rows[a]['Start_Timestamp'])
    for r in rows[a:b + 1]:
        o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {re.sub(r'[(<].*', '', r['Kernel_Name'])[:70]}\n")
PY
