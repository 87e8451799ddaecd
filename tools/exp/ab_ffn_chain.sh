#!/bin/bash
# A/B on one box: the bf16-mode JTA train step with the chained feed-forward kernels on / off (EMLOCO_FFN_CHAIN), interleaved,
# then a kernel trace of the chained build.
for rep in 1 2; do
  for c in 0 1; do
    EMLOCO_FFN_CHAIN=$c JTA_PRECISION=bf16 python tools/exp/jta_step.py 6 2>/dev/null | tail -1 | sed "s/^/chain=$c bf16: /"
  done
done
EMLOCO_FFN_CHAIN=1 JTA_PRECISION=fp32_split python tools/exp/jta_step.py 4 2>/dev/null | tail -1 | sed "s/^/fp32_split (chain not used): /"
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pj; JTA_PRECISION=bf16 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pj -- python $R/tools/exp/jta_step.py 4 > /tmp/pj.log 2>&1
python - "$(find /tmp/pj -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
steps = 6
for i,r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<24: print(f'{float(r["TotalDurationNs"])/1e6/steps:8.2f} ms/step {int(r["Calls"])/steps:6.1f} calls/step avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:100]}')
PY
