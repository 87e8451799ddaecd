#!/bin/bash
# A/B on one box: the LocoVal fit every step vs in groups of k steps (EMLOCO_FIT_EVERY), headline loop of bench.py, interleaved
for rep in 1 2 3; do
  for k in 1 4 8; do
    EMLOCO_FIT_EVERY=$k python bench.py --no_cpu_baseline --no_jta --no_policy --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fit_every=$k', d['value'], d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'env_only', d['env_step_only']['value'], d['env_step_only']['ms_per_step'])"
  done
done
# the driver's short window
for k in 1 4; do
  EMLOCO_FIT_EVERY=$k python bench.py --no_cpu_baseline --no_jta --no_policy --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps20 fit_every=$k', d['value'], d['ms_per_step'])"
done
