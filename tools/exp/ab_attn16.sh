#!/bin/bash
# A/B on one box: bf16-mode JTA train step on round 4's attention kernels (EMLOCO_ATTN16_OLD=1) vs round 5's (two blocks per wave, bf16
# tile images), interleaved; then per-kernel times and the matrix-pipe busy fraction of the new build.
for rep in 1 2; do
  for o in 1 0; do
    EMLOCO_ATTN16_OLD=$o JTA_PRECISION=bf16 python tools/exp/jta_step.py 6 2>/dev/null | tail -1 | sed "s/^/attn16_old=$o bf16: /"
  done
done
R=$PWD; cd /tmp; export TMPDIR=/tmp JTA_PRECISION=bf16
rm -rf /tmp/pa_kt /tmp/pa_mf
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa_kt -- python $R/tools/exp/jta_step.py 4 > /tmp/pa_kt.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pa_mf -- python $R/tools/exp/jta_step.py 4 > /tmp/pa_mf.log 2>&1
python - <<'PY'
import csv, glob, collections
steps = 6
avg = {}
print("per-kernel times, bf16-mode JTA step (rocprofv3 --kernel-trace --stats -- python tools/exp/jta_step.py 4; 2 warm-up + 4 steps):")
for f in glob.glob("/tmp/pa_kt/**/*kernel_stats.csv", recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        avg[r["Name"][:70]] = float(r["AverageNs"])
        if i < 16: print(f'{float(r["TotalDurationNs"])/1e6/steps:8.2f} ms/step {int(r["Calls"])/steps:6.1f} calls/step avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:100]}')
busy, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob("/tmp/pa_mf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            k = r["Kernel_Name"][:70]; busy[k] += float(r["Counter_Value"]); n[k] += 1
print("kernel | launches | MFMA busy cycles per launch (sum over the 1024 SIMDs) | avg duration [us] (kernel-trace pass) | MFMA utilisation = busy / (duration x 2.4 GHz x 1024)")
for k in sorted(busy, key=lambda k: -busy[k])[:12]:
    if busy[k] > 0 and k in avg:
        per = busy[k] / n[k]
        print(f"{k:70s} | {n[k]:5d} | {per:.4g} | {avg[k] / 1e3:9.1f} | {per / (avg[k] * 2.4 * 1024):.3f}")
PY
