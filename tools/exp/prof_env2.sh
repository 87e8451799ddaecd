#!/bin/bash
# one step of the DEFAULT (overlapped) schedule in launch order, with the queue of each kernel: bash tools/exp/prof_env2.sh <out.txt>
out=${1:-gpurun_out/env_trace2.txt}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_env2
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_env2 -- python $R/bench.py --steps 120 --warmup 20 --no_cpu_baseline --no_policy --no_jta > /tmp/prof_env2.log 2>&1
cd $R
python - "$out" <<'PY'
import csv, glob, sys, re
f = glob.glob('/tmp/prof_env2/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
def wgs(r):
    return int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) // max(int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 64)) or 64), 1)
big = [i for i, r in enumerate(rows) if 'sim_step_kernel' in r['Kernel_Name'] and wgs(r) > 4096]
with open(sys.argv[1], 'w') as o:
    for label, k in [(f"at {pc} % of the run's rigid-body launches", len(big) * pc // 100) for pc in (10, 35, 60, 85)]:
        a, b = big[k], big[k + 1]
        t0 = int(rows[a]['Start_Timestamp'])
        o.write(f"\n{label}: one step in launch order (start us, duration us, queue, workgroups, kernel)\n")
        for r in rows[a:b + 1]:
            o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} {wgs(r):6d}  {re.sub(r'[(<].*', '', r['Kernel_Name'])[:60]}\n")
print(open(sys.argv[1]).read())
PY
