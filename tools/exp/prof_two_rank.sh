#!/bin/bash
# Two ranks of bench.py sharing ONE GPU (EMLOCO_BENCH_SHARE_GPU=1: gloo stands in for RCCL, which refuses two ranks on one device):
# kernel + memory-copy trace of rank 0 around one step of the LocoVal loop, with queue ids -- where the per-step gradient exchange
# (6 176 floats) lands relative to sim_step_kernel.   bash tools/exp/prof_two_rank.sh <out.txt>
out=${1:-gpurun_out/two_rank_trace.txt}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_2r
EMLOCO_BENCH_SHARE_GPU=1 timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_2r -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 $R/bench.py --gpus 2 --steps 60 --warmup 20 \
  --num_envs 2048 --no_jta --no_policy --no_cpu_baseline > /tmp/prof_2r.log 2>&1
cd $R
python - "$out" <<'PY'
import csv, glob, sys, re, collections
files = sorted(glob.glob('/tmp/prof_2r/**/*kernel_trace.csv', recursive=True))
best = None
for f in files:                       # the rank processes are the ones that launched sim_step_kernel; take the first
    rows = list(csv.DictReader(open(f)))
    if any('sim_step_kernel' in r['Kernel_Name'] for r in rows):
        best = (f, rows); break
with open(sys.argv[1], 'w') as o:
    if best is None:
        o.write("no kernel trace with sim_step_kernel found: " + " ".join(files) + "\n" + open('/tmp/prof_2r.log').read()[-2000:])
        sys.exit(0)
    f, rows = best
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'q' + str(r.get('Queue_Id', '?')), re.sub(r'[(<].*', '', r['Kernel_Name'])[:50]) for r in rows]
    mc = f.replace('kernel_trace', 'memory_copy_trace')
    try:
        for r in csv.DictReader(open(mc)):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy', r.get('Direction', r.get('Name', 'memcpy'))[:50]))
    except Exception as e:
        o.write(f"(no memory-copy trace: {e})\n")
    ev.sort()
    big = [i for i, e in enumerate(ev) if 'sim_step_kernel' in e[3]]
    o.write(f"rank process trace {f.split('/')[-1]}; two ranks x 2048 envs on one GPU (gloo test mode), LocoVal loop; {len(big)} rigid-body launches\n")
    for pc in (78, 92):
        a, b = big[len(big) * pc // 100], big[len(big) * pc // 100 + 1]
        t0 = ev[a][0]
        o.write(f"\none step at {pc} % of the run in start order (start us, duration us, queue | copy, what)\n")
        for s, e, q, n in ev[a:b + 1]:
            o.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {q:>5}  {n}\n")
print(open(sys.argv[1]).read()[:6000])
PY
