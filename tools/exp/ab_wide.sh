#!/bin/bash
# A/B on one box: JTA train step with the GEMMs' LDS-transposed 16-byte epilogue stores off / on (EMLOCO_GEMM_WIDE_STORES), both precisions, interleaved
for rep in 1 2; do
  for p in fp32_split bf16; do
    for w in 0 1; do
      EMLOCO_GEMM_WIDE_STORES=$w JTA_PRECISION=$p python tools/exp/jta_step.py 5 2>/dev/null | tail -1 | sed "s/^/wide=$w $p: /"
    done
  done
done
