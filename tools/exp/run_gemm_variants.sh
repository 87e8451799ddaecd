#!/bin/bash
# On the GPU box: tools/exp/probe_split.py per variant built by mk_gemm_variant.sh.   Usage: bash tools/exp/run_gemm_variants.sh NAME...
LIB=emloco_amd/lib/libemloco_hip.so
cp $LIB /tmp/orig.so
for N in "$@"; do
  echo "=== $N"
  cp variants/$N.so $LIB
  timeout 300 python tools/exp/probe_split.py ${PROBE_ARGS} 2>&1 | grep -v amdgpu.ids
done
cp /tmp/orig.so $LIB
