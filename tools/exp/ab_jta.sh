#!/bin/bash
# A/B two builds of libemloco_hip.so on the JTA train step (fp32-class and bf16) and the chain / policy legs: bash tools/exp/ab_jta.sh A.so B.so
A=$1; B=$2
LIB=emloco_amd/lib/libemloco_hip.so
cp $LIB /tmp/orig.so
for rep in 1 2; do
  for v in A B; do
    f=$A; [ $v = B ] && f=$B
    cp $f $LIB
    echo "$v fp32-class: $(python tools/exp/jta_step.py 4 2>/dev/null | tail -1)   bf16: $(JTA_PRECISION=bf16 python tools/exp/jta_step.py 4 2>/dev/null | tail -1)"
    python bench.py --no_cpu_baseline --no_jta --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], 'env_only', d['env_step_only']['value'], 'policy', d['policy']['value'], d['policy']['policy_ms'])"
  done
done
cp /tmp/orig.so $LIB
