#!/bin/bash
# Timing loop for sim_step_kernel experiments (run via gpurun): per-phase stamps of env 0, then the kernel's HIP-event time in
# the rollout bench.  Usage: bash tools/exp/sim_time.sh [extra hipcc flags]
X="$*"
EMLOCO_HIPCC_EXTRA="-DEMLOCO_SIM_PROFILE=1 $X" python -m emloco_amd.build --force > /dev/null 2>&1 || { echo "profile build failed"; exit 1; }
timeout 300 python tools/sim_phase_profile.py 4096 2>&1 | tail -12
EMLOCO_HIPCC_EXTRA="$X" python -m emloco_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for i in 1 2; do
timeout 600 python bench.py --no_jta --no_policy --no_cpu_baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rollout', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'env_only', d['env_step_only']['value'])"
done
