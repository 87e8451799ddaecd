#!/bin/bash
# A/B on one box: fp32-class JTA train step on round 4's split-mode attention kernels (EMLOCO_ATTN16_OLD=1) vs round 5's (piece-plane tile images), interleaved,
# and the kernels alone (tools/exp/attn_probe.py, ATTN_MODE=split)
for rep in 1 2; do
  for o in 1 0; do
    EMLOCO_ATTN16_OLD=$o JTA_PRECISION=fp32_split python tools/exp/jta_step.py 5 2>/dev/null | tail -1 | sed "s/^/attn16_old=$o fp32_split: /"
  done
done
ATTN_MODE=split EMLOCO_ATTN16_OLD=1 python tools/exp/attn_probe.py variants/attn_cur.so 2>&1 | grep -v amdgpu | sed "s/^/round-4 kernels: /"
ATTN_MODE=split python tools/exp/attn_probe.py variants/attn_cur.so 2>&1 | grep -v amdgpu | sed "s/^/round-5 kernels: /"
