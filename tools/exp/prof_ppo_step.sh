#!/bin/bash
# ONE graphed PPO + AMP optimiser step in START order with queue ids (which arm runs where, what the critical path is):
#   bash tools/exp/prof_ppo_step.sh <out.txt>      (GPU box; rocprofv3 --kernel-trace of tools/exp/ppo_epoch.py)
out=${1:-gpurun_out/r05/ppo_step_trace.txt}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ppo_step
timeout 1200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ppo_step -- python $R/tools/exp/ppo_epoch.py 1 > /tmp/prof_ppo_step.log 2>&1
cd $R
python - "$out" <<'PY'
import csv, glob, sys, re, collections
rows = []
for f in glob.glob('/tmp/prof_ppo_step/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Queue_Id", 0) or 0), re.sub(r"\(.*", "", r["Kernel_Name"])[:90],
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 64)) or 64), 1)))
rows.sort()
# optimiser steps end with the fused clip + Adam launch: take the window between two consecutive ones, three quarters into the run
marks = [i for i, r in enumerate(rows) if "adam" in r[3].lower() and "clip" in r[3].lower()] or [i for i, r in enumerate(rows) if "adam" in r[3].lower()]
with open(sys.argv[1], "w") as o:
    o.write("one graphed PPO + AMP optimiser step (4096 envs x horizon 32, minibatch 2048) in START order: start us, duration us, queue, workgroups, kernel\n")
    o.write("".join(l for l in open('/tmp/prof_ppo_step.log') if l.startswith("{"))[:600] + "\n")
    if len(marks) < 8:
        o.write("no optimiser-step marker found; kernels: %s\n" % collections.Counter(r[3] for r in rows).most_common(20))
        sys.exit(0)
    # the last launch of a step = the last marker of a burst; find bursts
    ends = [m for k, m in enumerate(marks) if k + 1 == len(marks) or rows[marks[k + 1]][0] - rows[m][1] > 200000]
    a, b = ends[len(ends) * 3 // 4 - 1], ends[len(ends) * 3 // 4]
    win = rows[a + 1:b + 1]
    t0 = win[0][0]
    busy = collections.defaultdict(float)
    for s, e, q, n, g in win:
        o.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q:<3d} {g:6d}  {n}\n")
        busy[q] += (e - s) / 1e3
    o.write(f"step: {(win[-1][1] - t0) / 1e3:.1f} us, {len(win)} launches; busy us per queue: " + ", ".join(f"q{q}: {v:.0f}" for q, v in sorted(busy.items())) + "\n")
    small = sum(1 for s, e, q, n, g in win if e - s < 8000)
    o.write(f"launches under 8 us: {small} ({sum((e - s) / 1e3 for s, e, q, n, g in win if e - s < 8000):.0f} us of kernel time)\n")
PY
tail -5 $out
