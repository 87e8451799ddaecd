#!/bin/bash
# HBM counters of sim_step_kernel per launch for given split settings (run on the GPU box): bash tools/exp/traffic.sh 2 4
R=$(pwd); export TMPDIR=/tmp EMLOCO_OVERLAP_RESET=0
for SP in "$@"; do
  export EMLOCO_SPLIT=$SP
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/tr_$C && (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d /tmp/tr_$C -- python $R/bench.py --steps 20 --warmup 5 --no_jta --no_policy --no_cpu_baseline > /tmp/tr_$C.log 2>&1)
  done
  python - $SP <<'PY'
import csv, glob, sys
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"/tmp/tr_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C and "sim_step_kernel" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    res[C] = sum(v) / max(len(v), 1)
rd, wr = 2 * res["FETCH_SIZE"] * 1024 / 1e6, res["WRITE_SIZE"] * 1024 / 1e6
print(f"split {sys.argv[1]}: read {rd:.1f} MB (FETCH x2) + write {wr:.1f} MB = {rd + wr:.1f} MB per launch")
PY
done
