#!/bin/bash
# per-kernel time of the PPO + AMP train_epoch (bench.py `policy.ppo`): bash tools/exp/prof_ppo.sh <out.txt> [epochs]   (GPU box)
out=${1:-gpurun_out/r04/ppo_kernels.txt}; epochs=${2:-1}
R=$(pwd); mkdir -p $(dirname $out)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ppo
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ppo -- python $R/tools/exp/ppo_epoch.py $epochs > /tmp/prof_ppo.log 2>&1
cd $R
python - "$out" "$epochs" <<'PY'
import csv, glob, sys, re
out, epochs = sys.argv[1], int(sys.argv[2]) + 1
f = glob.glob('/tmp/prof_ppo/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = 0.0
with open(out, 'w') as o:
    o.write("rocprofv3 --kernel-trace --stats -- python tools/exp/ppo_epoch.py: AMPAgent.train_epoch, 4096 envs x horizon 32, warm-up epoch included in the averages\n")
    o.write(open('/tmp/prof_ppo.log').read()[-700:] + "\n")
    for r in rows[:45]:
        ms = float(r['TotalDurationNs']) / 1e6 / epochs
        tot += ms
        name = re.sub(r'\(.*', '', r['Name'])[:110]
        o.write(f"{ms:9.2f} ms/epoch  {int(r['Calls'])//epochs:7d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {name}\n")
    o.write(f"total (top 45) {tot:.1f} ms/epoch\n")
PY
cat $out
