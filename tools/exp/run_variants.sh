#!/bin/bash
# On the GPU box: per variant built by mk_variant.sh, the per-phase stamps of env 0 and the rollout leg's kernel time (2 runs).
# Usage: bash tools/exp/run_variants.sh NAME...
LIB=emloco_amd/lib/libemloco_hip.so
cp $LIB /tmp/orig.so
for N in "$@"; do
  echo "=== $N"
  cp variants/${N}_prof.so $LIB
  timeout 300 python tools/sim_phase_profile.py 4096 2>&1 | tail -15
  cp variants/${N}.so $LIB
  for i in 1 2; do
    timeout 600 python bench.py --no_jta --no_policy --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rollout', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'env_only', d['env_step_only']['value'])"
  done
done
cp /tmp/orig.so $LIB
