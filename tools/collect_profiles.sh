#!/bin/bash
# Collect the rocprofv3 evidence for one round on the MI355X box.  Usage (from the repo root, via gpurun):
#   bash tools/collect_profiles.sh r01
# Writes small summaries under gpurun_out/<round>/ (copy the ones to be judged into profiles/).
# Kernel-trace statistics and the PMC counters are collected in SEPARATE runs (FETCH_SIZE and WRITE_SIZE do not
# fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -u
R=${1:-r01}
OUT=$PWD/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_ppo"
T="timeout 900"      # a rocprofv3 pass that hangs must not eat the GPU budget

# 1. kernel trace + stats of the bench command (env leg + JTA leg)
rm -rf /tmp/prof_kt && (cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- $BENCH > "$OUT/bench_under_rocprof.log" 2>&1)
f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && python - "$f" "$OUT/${R}_bench_kernel_stats_top.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as o:
    w = csv.writer(o)
    for r in rows[:31]:
        r[0] = r[0][:110]
        w.writerow(r)
PY

# 1b. the rigid-body launches of pass 1 by kind: the default schedule issues two per step (live envs: grid = 2 workgroups per
#     env; the reset envs' id-list launch), the `sequential` leg one -- the stats row above averages over all of them
f=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python - "$f" "$OUT/${R}_sim_step_launches.txt" <<'PY'
import csv, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "sim_step_kernel" in r["Kernel_Name"]:
        wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 64)) or 64)
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        g[grid // max(wg, 1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(sys.argv[2], "w") as o:
    o.write("sim_step_kernel launches of `bench.py --steps 100 --warmup 10 --no_cpu_baseline` by grid (workgroups): calls, avg / min / max us\n")
    for k in sorted(g, reverse=True):
        v = g[k]
        o.write(f"  {k:6d} workgroups: {len(v):5d} calls  avg {sum(v)/len(v):8.1f}  min {min(v):8.1f}  max {max(v):8.1f}\n")
print(open(sys.argv[2]).read())
PY

# 2. PMC passes (env leg only), one counter per pass.  EMLOCO_OVERLAP_RESET=0: the sequential schedule -- one rigid-body launch
#    per step for all envs -- so that "per launch" below is the whole step's kernel (the counters do not depend on the schedule)
export EMLOCO_OVERLAP_RESET=0
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C && (cd /tmp && $T rocprofv3 --pmc $C --output-format csv -d /tmp/prof_$C -- $BENCH --steps 20 --warmup 5 --no_jta --no_policy > "$OUT/pmc_$C.log" 2>&1)
done
python - "$OUT/${R}_pmc_summary.txt" "$OUT/${R}_sim_step_hbm_bytes.json" <<'PY'
import csv, glob, json, sys, collections
res = {}
lines = []
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/prof_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        lines.append(f"{C} {k:60s} launches {len(v):5d} mean_KB {sum(v)/len(v):14.2f}")
        if "sim_step_kernel" in k:
            res[C] = sum(v) / len(v)
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
if len(res) == 2:
    # rocprofv3 reports KB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM)
    rd, wr = 2.0 * res["FETCH_SIZE"] * 1024.0, res["WRITE_SIZE"] * 1024.0
    json.dump({"kernel": "sim_step_kernel", "num_envs": 4096, "fetch_size_kb_raw": res["FETCH_SIZE"],
               "write_size_kb_raw": res["WRITE_SIZE"], "read_bytes_corrected": rd, "write_bytes": wr,
               "hbm_bytes_per_launch": rd + wr,
               "note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported; separate --pmc passes"},
              open(sys.argv[2], "w"), indent=1)
print("\n".join(lines))
PY
# 3. MFMA utilisation of the predictor kernels (JTA leg): busy cycles of the matrix pipes over all SIMDs vs GPU-active cycles
JTA="$BENCH --steps 10 --warmup 2 --no_policy"      # the same command for the duration pass and the counter pass
rm -rf /tmp/prof_mfma_kt && (cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mfma_kt -- $JTA > "$OUT/mfma_kt.log" 2>&1)
rm -rf /tmp/prof_mfma && (cd /tmp && $T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/prof_mfma -- $JTA > "$OUT/pmc_mfma.log" 2>&1)
python - "$OUT/${R}_mfma_utilisation.txt" "$(find /tmp/prof_mfma_kt -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, glob, sys, collections
busy, n = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob("/tmp/prof_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            k = r["Kernel_Name"][:70]
            busy[k] += float(r["Counter_Value"]); n[k] += 1
avg_ns = {}
for r in list(csv.reader(open(sys.argv[2])))[1:]:
    avg_ns[r[0][:70]] = float(r[3])
CLK = 2.4          # GHz, MI355X peak engine clock (MI355X_MICROARCH.md); 256 CUs x 4 SIMDs
lines = ["kernel | launches | MFMA busy cycles per launch (sum over the 1024 SIMDs) | avg duration [us] (kernel-trace pass) | "
         "MFMA utilisation = busy / (duration x 2.4 GHz x 1024)"]
for k in sorted(busy, key=lambda k: -busy[k])[:20]:
    if busy[k] > 0 and k in avg_ns:
        per = busy[k] / n[k]
        lines.append(f"{k:70s} | {n[k]:5d} | {per:.4g} | {avg_ns[k] / 1e3:9.1f} | {per / (avg_ns[k] * CLK * 1024):.3f}")
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
# 4. VALU occupancy of the rollout kernel (env leg): where the wave cycles of sim_step_kernel go.  SQ counters only (8 slots),
#    its own pass; the duration comes from a kernel-trace pass of the same (sequential-schedule) command.
rm -rf /tmp/prof_kt_seq && (cd /tmp && $T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt_seq -- $BENCH --steps 20 --warmup 5 --no_jta --no_policy > "$OUT/kt_seq.log" 2>&1)
rm -rf /tmp/prof_valu && (cd /tmp && $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_valu -- $BENCH --steps 20 --warmup 5 --no_jta --no_policy > "$OUT/pmc_valu.log" 2>&1)
python - "$OUT/${R}_sim_step_valu.txt" "$(find /tmp/prof_kt_seq -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/prof_valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sim_step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
dur_ns = None
for r in list(csv.reader(open(sys.argv[2])))[1:]:
    if "sim_step_kernel" in r[0]:
        dur_ns = float(r[3])
lines = ["sim_step_kernel, 4096 envs (one 64-lane wave per env and substep: 4 dependent workgroups per env), sequential schedule, mean per launch over %d launches" % len(next(iter(agg.values()), []))]
for k in sorted(m):
    lines.append(f"  {k:22s} {m[k]:.6g}")
if m.get("SQ_WAVE_CYCLES"):
    wc = m["SQ_WAVE_CYCLES"]
    lines.append("fractions of the wave cycles (SQ_WAVE_CYCLES; quad-cycle units cancel):")
    for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in m:
            lines.append(f"  {k:22s} {m[k] / wc:.3f}")
if dur_ns and m.get("SQ_INSTS_VALU"):
    CLK, SIMDS = 2.4, 1024          # GHz nominal engine clock, 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
    cyc = dur_ns * CLK
    lines.append(f"duration (kernel-trace pass) {dur_ns / 1e3:.1f} us = {cyc:.4g} cycles at {CLK} GHz")
    lines.append(f"VALU wave-instructions per launch {m['SQ_INSTS_VALU']:.4g} = {m['SQ_INSTS_VALU'] / 4096:.0f} per env")
    lines.append(f"VALU issue utilisation = insts x 2 cycles (wave64 on a SIMD-32) / (cycles x {SIMDS} SIMDs) = "
                 f"{m['SQ_INSTS_VALU'] * 2 / (cyc * SIMDS):.3f}  (lower bound: the engine clock under load is below nominal)")
    if m.get("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles in which a wave has a VALU instruction in flight: x 4 = cycles, summed over the waves
        # of a SIMD; measured cost per instruction (active / insts) and the share of the SIMD-cycles of the launch it covers
        per = m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_INSTS_VALU"]
        lines.append(f"VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (cycles x {SIMDS} SIMDs) = {m['SQ_ACTIVE_INST_VALU'] * 4 / (cyc * SIMDS):.3f}"
                     f"  ({per:.2f} cycles of VALU-active time per instruction)")
    if m.get("SQ_WAVE_CYCLES"):
        lines.append(f"resident waves per SIMD (time average) = SQ_WAVE_CYCLES x 4 / (cycles x {SIMDS}) = {m['SQ_WAVE_CYCLES'] * 4 / (cyc * SIMDS):.2f}")
open(sys.argv[1], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
echo "collected into $OUT"; ls -la "$OUT"
