#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them for kernels that move exactly 1 GiB in and 1 GiB out (separate passes).
export TMPDIR=/tmp
R=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/cal_$C -- $R/tools/exp/_build/pmc_calib > /tmp/cal_$C.log 2>&1)
  python - $C <<'PY'
import csv, glob, sys, collections
C = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/cal_{C}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == C:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    kb = sum(v) / len(v)
    print(f"{C:11s} {k:16s} reported {kb:14.1f} (KB)  = {kb * 1024 / 2**30:.4f} x the 1 GiB moved   -> factor to bytes moved: {2**30 / (kb * 1024):.3f}")
PY
done
