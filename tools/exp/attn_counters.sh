#!/bin/bash
# where the wave cycles of the fused attention kernels go: SQ counters of one JTA train step: bash tools/exp/attn_counters.sh <label> [fp32_split|bf16]
L=${1:-x}; P=${2:-fp32_split}
R=$PWD; OUT=$R/gpurun_out/r05; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp JTA_PRECISION=$P
rm -rf /tmp/ac1 /tmp/ac2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/ac1 -- python $R/tools/exp/jta_step.py 2 > /tmp/ac1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d /tmp/ac2 -- python $R/tools/exp/jta_step.py 2 > /tmp/ac2.log 2>&1
python - $P > $OUT/attn_counters_${L}_${P}.txt <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/ac1", "/tmp/ac2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if any(t in k for t in (os.environ.get("KFILTER") or "attn,ffn_chain").split(",")):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"SQ counters of the attention kernels, JTA train step ({sys.argv[1]}), rocprofv3 --pmc (two passes), the LARGEST launches of each kernel (the five full local layers), mean per launch")
for k, c in agg.items():
    # keep the big launches: those within 2x of the max wave cycles
    wc = c.get("SQ_WAVE_CYCLES", [])
    print(f"\n{k}")
    def big(v):
        m = max(v) if v else 0
        sel = [x for x in v if x > 0.5 * m]
        return sum(sel) / max(len(sel), 1)
    vals = {n: big(v) for n, v in c.items()}
    for n in sorted(vals): print(f"   {n:26s} {vals[n]:.5g}")
    if vals.get("SQ_WAVE_CYCLES"):
        w = vals["SQ_WAVE_CYCLES"]
        print("   fractions of the wave cycles: " + "  ".join(f"{n[3:]} {vals[n] / w:.3f}" for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS") if n in vals))
        if vals.get("SQ_BUSY_CYCLES"): print(f"   waves resident per SIMD (time average) ~ WAVE_CYCLES / BUSY_CYCLES x ... : {w / vals['SQ_BUSY_CYCLES']:.2f} (per SQ)")
PY
cat $OUT/attn_counters_${L}_${P}.txt
