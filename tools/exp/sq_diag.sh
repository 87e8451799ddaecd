#!/bin/bash
# On the GPU box: SQ / instruction-cache / LDS counters of sim_step_kernel on the bench workload, several --pmc passes
# (counters only; separate passes).  Usage: bash tools/exp/sq_diag.sh OUTFILE
export TMPDIR=/tmp
OUT=${1:-gpurun_out/sq_diag.txt}
CMD="python $PWD/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_jta --no_policy"
export EMLOCO_OVERLAP_RESET=0
SETS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_MFMA"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
      "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL"
      "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"
      "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
      "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY")
: > $OUT
i=0
for S in "${SETS[@]}"; do
  rm -rf /tmp/sqd_$i
  (cd /tmp && timeout 600 rocprofv3 --pmc $S --output-format csv -d /tmp/sqd_$i -- $CMD > /tmp/sqd_$i.log 2>&1)
  python - /tmp/sqd_$i >> $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sim_step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} {sum(v)/len(v):16.6g}   (launches {len(v)})")
PY
  i=$((i+1))
done
cat $OUT
