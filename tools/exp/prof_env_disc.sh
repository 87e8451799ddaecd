#!/bin/bash
# one step of the configs[2] loop in launch order with the queue of every kernel: bash tools/exp/prof_env_disc.sh <out.txt>
out=${1:-gpurun_out/r04/env_step_trace_disc.txt}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_envd
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_envd -- python $R/tools/exp/env_disc_loop.py 120 > /tmp/prof_envd.log 2>&1
cd $R
python - "$out" <<'PY'
import csv, glob, sys, re
f = glob.glob('/tmp/prof_envd/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
def wgs(r):
    return int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) // max(int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 64)) or 64), 1)
big = [i for i, r in enumerate(rows) if 'sim_step_kernel' in r['Kernel_Name'] and wgs(r) > 4096]
with open(sys.argv[1], 'w') as o:
    o.write("configs[2] loop (frozen policy + AMP discriminator + LocoVal fit, 4096 envs), rocprofv3 --kernel-trace: one step in START order;\n"
            "columns: start us (from the rigid-body launch), duration us, queue, workgroups, kernel.  The policy / reset chain / rigid-body launch\n"
            "share one queue (the caller's stream); the discriminator's GEMMs, the return bookkeeping's second half and the fit are on another.\n")
    o.write(open('/tmp/prof_envd.log').read()[-400:] + "\n")
    for label, k in [(f"at {pc} % of the run's rigid-body launches", len(big) * pc // 100) for pc in (35, 70)]:
        a, b = big[k], big[k + 1]
        t0 = int(rows[a]['Start_Timestamp'])
        o.write(f"\n{label}:\n")
        for r in rows[a:b + 1]:
            o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} {wgs(r):6d}  {re.sub(r'[(<].*', '', r['Kernel_Name'])[:60]}\n")
print(open(sys.argv[1]).read()[-3000:])
PY
