#!/bin/bash
# A/B of the configs[2] loop's schedule knobs on one box (run through gpurun): prints the `with_discriminator_and_locoval_fit` line per variant
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/ab_disc.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no_cpu_baseline --no_jta --no_ppo 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); p = d['policy']; w = p['with_discriminator_and_locoval_fit']
        print('headline', d['value'], 'policy', p['value'], p['policy_ms'], 'disc+fit', w['value'], w['ms_per_step'])
" >> $OUT
}
run A=1
run EMLOCO_DISC_UNDER_PHYSICS=0
run EMLOCO_KSPLIT_TARGET=768
run EMLOCO_FIT_PRIORITY=1
run EMLOCO_FIT_PRIORITY=-1
run EMLOCO_KSPLIT_TARGET=768 EMLOCO_FIT_PRIORITY=1
run EMLOCO_DEFER_DISC=0
cat $OUT
