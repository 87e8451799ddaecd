#!/bin/bash
# On the GPU box: sim_step_kernel time against the number of resident workgroups per CU (EMLOCO_SIM_LDS_PAD reserves extra
# dynamic LDS per workgroup: 0 -> 8 per CU (register-limited), 8192 -> 6, 17408 -> 4, 40000 -> 2).  Latency-bound: time ~ 1 / residency.
for pad in 0 8192 17408 40000; do
  echo "=== pad $pad"
  EMLOCO_SIM_LDS_PAD=$pad timeout 600 python bench.py --no_jta --no_policy --no_cpu_baseline --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('rollout', d['value'], 'ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'seq', d.get('sequential', {}).get('kernel_ms'), d.get('sequential', {}).get('value'))"
done
