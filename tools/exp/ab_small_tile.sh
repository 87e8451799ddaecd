#!/bin/bash
# A/B of the 64 x 64 split tile and the split-K target on the policy / discriminator / PPO legs (one box; run through gpurun)
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/ab_small_tile.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_jta --no_ppo 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['policy']; w = p['with_discriminator_and_locoval_fit']
        print('headline', d['value'], 'policy', p['value'], 'policy_ms', p['policy_ms'], 'tflops', p['roofline']['achieved'], 'disc+fit', w['value'], w['ms_per_step'])
" >> $OUT
}
run EMLOCO_GEMM_SMALL=0
run A=1
run EMLOCO_KSPLIT_TARGET=256
run EMLOCO_KSPLIT_TARGET=128
run EMLOCO_KSPLIT_TARGET=1024
run EMLOCO_GEMM_SMALL=1
cat $OUT
