#!/bin/bash
# A/B of the split-mode attention on the JTA train step (one box)
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/ab_attn_split.txt
: > $OUT
for v in 0 1; do
  echo "== EMLOCO_ATTN_SPLIT=$v" >> $OUT
  EMLOCO_ATTN_SPLIT=$v timeout 400 python tools/exp/jta_time.py 2>/dev/null | tail -3 >> $OUT
done
cat $OUT
