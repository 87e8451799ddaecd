#!/bin/bash
# HBM traffic of the JTA train step's kernels (on the GPU box): bash tools/exp/prof_jta_hbm.sh <label> [fp32_split|bf16]
# Three passes of tools/exp/jta_step.py: kernel trace (durations), --pmc FETCH_SIZE, --pmc WRITE_SIZE (they do not fit one pass;
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 B, same guide).
# Output: per kernel -- launches per step, average duration, read / written bytes per launch, effective TB/s.
L=${1:-x}; P=${2:-fp32_split}
R=$PWD; OUT=$R/gpurun_out/r05; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp JTA_PRECISION=$P JH_OUT=$OUT
STEPS=3
rm -rf /tmp/jh_kt /tmp/jh_F /tmp/jh_W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/jh_kt -- python $R/tools/exp/jta_step.py $STEPS > /tmp/jh_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/jh_F -- python $R/tools/exp/jta_step.py $STEPS > /tmp/jh_F.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/jh_W -- python $R/tools/exp/jta_step.py $STEPS > /tmp/jh_W.log 2>&1
python - $STEPS $P > $OUT/jta_hbm_${L}_${P}.txt <<'PY'
import csv, glob, sys, collections
steps, prec = int(sys.argv[1]), sys.argv[2]
calls = steps + 2          # jta_step.py: 2 warm-up steps + `steps` timed
dur = {}
for f in glob.glob("/tmp/jh_kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"][:80]] = (int(r["Calls"]), float(r["AverageNs"]))
cnt = {}
for C, d in (("FETCH_SIZE", "/tmp/jh_F"), ("WRITE_SIZE", "/tmp/jh_W")):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                agg[r["Kernel_Name"][:80]].append(float(r["Counter_Value"]))
    cnt[C] = {k: sum(v) / len(v) for k, v in agg.items()}
print(f"JTA EmLoco train step, B = 256, precision {prec}: HBM traffic per kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes;")
print("read = FETCH_SIZE x 2 x 1024 B (gfx950 correction), written = WRITE_SIZE x 1024 B; durations from a --kernel-trace --stats pass of the same command)")
print(f"{'kernel':80s} {'calls/step':>10s} {'avg us':>9s} {'read MB':>9s} {'write MB':>9s} {'TB/s':>6s} {'ms/step':>8s} {'GB/step':>8s}")
rows = []
for k, (n, ns) in dur.items():
    rd = 2.0 * cnt["FETCH_SIZE"].get(k, 0.0) * 1024.0
    wr = cnt["WRITE_SIZE"].get(k, 0.0) * 1024.0
    rows.append((n * ns, k, n, ns, rd, wr))
tot_ms = tot_gb = 0.0
for t, k, n, ns, rd, wr in sorted(rows, reverse=True)[:28]:
    tbs = (rd + wr) / ns / 1e3 if ns else 0.0
    print(f"{k:80s} {n / calls:10.1f} {ns / 1e3:9.1f} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {tbs:6.2f} {t / calls / 1e6:8.2f} {(rd + wr) * n / calls / 1e9:8.2f}")
    tot_ms += t / calls / 1e6; tot_gb += (rd + wr) * n / calls / 1e9
print(f"top rows: {tot_ms:.1f} ms and {tot_gb:.1f} GB per step -> {tot_gb / tot_ms:.2f} TB/s on average while a kernel runs")
import json, os
gem = [(n, ns, rd, wr) for t, k, n, ns, rd, wr in rows if "gemm" in k]
gb = sum((rd + wr) * n for n, ns, rd, wr in gem) / calls
gms = sum(n * ns for n, ns, rd, wr in gem) / calls / 1e6
json.dump({"precision": prec, "gemm_bytes_per_step": gb, "gemm_ms_per_step": gms, "gemm_tb_per_s": gb / gms / 1e9 if gms else None,
           "note": "all kernels whose name contains 'gemm' (split-K reductions included): FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per launch x launches per step; "
                   "separate --pmc passes of tools/exp/jta_step.py; FETCH counts fabric requests (Infinity-Cache hits included)"},
          open(os.path.join(os.environ.get("JH_OUT", "/tmp"), f"jta_gemm_hbm_bytes_{prec}.json"), "w"), indent=1)
PY
cat $OUT/jta_hbm_${L}_${P}.txt; tail -2 /tmp/jh_kt.log
