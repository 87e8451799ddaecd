#!/bin/bash
# every distinct (kernel, grid) of one JTA train step with its count and average duration: bash tools/exp/prof_jta_launches.sh <label> [fp32_split|bf16]
L=${1:-x}; P=${2:-fp32_split}
R=$PWD; OUT=$R/gpurun_out/r05; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp JTA_PRECISION=$P
rm -rf /tmp/jl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/jl -- python $R/tools/exp/jta_step.py 3 > /tmp/jl.log 2>&1
python - $P > $OUT/jta_launches_${L}_${P}.txt <<'PY'
import csv, glob, sys, collections, re
steps = 5      # 2 warm-up + 3
g = collections.defaultdict(list)
for f in glob.glob("/tmp/jl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        wg = max(int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 64)) or 64), 1)
        gx, gy, gz = (int(r.get(k, 1) or 1) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else (int(r.get("Grid_Size", 0) or 0), 1, 1)
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        g[(name, gx // (wg if "Grid_Size_X" in r else wg), gy, gz)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(((sum(v) / steps, k, len(v) / steps, sum(v) / len(v)) for k, v in g.items()), reverse=True)
print(f"JTA train step ({sys.argv[1]}): launches by (kernel, grid in workgroups x, y, z): ms per step, launches per step, average us")
tot = 0.0
for t, k, n, avg in rows[:60]:
    print(f"{t / 1e3:8.3f} ms  {n:6.1f} x {avg:9.1f} us  grid {k[1]:>6} {k[2]:>6} {k[3]:>4}  {k[0]}")
    tot += t / 1e3
print(f"listed: {tot:.1f} ms per step")
PY
cat $OUT/jta_launches_${L}_${P}.txt | head -70; tail -2 /tmp/jl.log
