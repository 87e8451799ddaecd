#!/bin/bash
# the two-half-batch pipeline with the policy in the loop (tools/exp/pipe_policy_probe.py, 2 shards of 2048 envs on 2 streams) under
# rocprofv3 --kernel-trace: one step of shard A in START order with the queue of every kernel: bash tools/exp/prof_pipe_policy.sh <out.txt>
out=${1:-gpurun_out/r05/env_step_trace_two_halves.txt}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_pp
PIPE_CONFIGS="2:0" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_pp -- python $R/tools/exp/pipe_policy_probe.py 100 > /tmp/prof_pp.log 2>&1
cd $R
python - "$out" <<'PY'
import csv, glob, sys, re
f = glob.glob('/tmp/prof_pp/**/*kernel_trace.csv', recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r['Start_Timestamp']))
def wgs(r):
    return int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) // max(int(r.get('Workgroup_Size', r.get('Workgroup_Size_X', 64)) or 64), 1)
big = [i for i, r in enumerate(rows) if 'sim_step_kernel' in r['Kernel_Name'] and wgs(r) >= 8192]
qs = sorted({rows[i].get('Queue_Id', '?') for i in big})
with open(sys.argv[1], 'w') as o:
    o.write("Two half-batches of 2048 envs on two streams, frozen policy in the loop (reset_done -> policy -> env.step per half; tools/exp/pipe_policy_probe.py\n"
            "PIPE_CONFIGS=2:0), rocprofv3 --kernel-trace: the kernels between two rigid-body launches of ONE half, in START order;\n"
            "columns: start us (from that half's rigid-body launch), duration us, queue, workgroups, kernel.  Queues of the rigid-body launches: " + ", ".join(qs) + "\n")
    o.write(open('/tmp/prof_pp.log').read()[-300:] + "\n")
    qa = rows[big[-3]].get('Queue_Id', '?')            # a stream of the timed loop (the pre-roll ran on the default stream's queue)
    mine = [i for i in big if rows[i].get('Queue_Id', '?') == qa]
    for label, k in [(f"at {pc} % of the run (the timed loop: the pre-roll without a policy is the first ~70 %)", len(mine) * pc // 100) for pc in (80, 92)]:
        a, b = mine[k], mine[k + 1]
        t0 = int(rows[a]['Start_Timestamp'])
        o.write(f"\n{label} (half on queue {qa}):\n")
        for r in rows[a:b + 1]:
            if int(r['End_Timestamp']) < t0: continue
            o.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3} {wgs(r):6d}  {re.sub(r'[(<].*', '', r['Kernel_Name'])[:60]}\n")
print(open(sys.argv[1]).read()[-5000:])
PY
