#!/bin/bash
# On the GPU box: the rollout leg of bench.py once per variants/*.so given (kernel_ms, env.step alone), then bit-exactness of the sim tests.
LIB=emloco_amd/lib/libemloco_hip.so
cp $LIB /tmp/orig.so
for N in "$@"; do
  cp variants/$N.so $LIB
  r=$(timeout 300 python bench.py --no_jta --no_policy --no_cpu_baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'env_only', d['env_step_only']['value'])" 2>&1 | tail -1)
  t=$(timeout 300 python -m pytest tests/test_gpu_sim.py -q -x 2>&1 | tail -1)
  echo "$N: $r | $t"
done
cp /tmp/orig.so $LIB
