#!/bin/bash
# A/B two builds of libemloco_hip.so on the same MI355X box (run via gpurun).  Usage: bash tools/ab_sim.sh A.so B.so [bench args]
# Each library is copied over emloco_amd/lib/libemloco_hip.so in turn and the rollout leg of bench.py is run 3 times, interleaved.
A=$1; B=$2; shift 2
LIB=emloco_amd/lib/libemloco_hip.so
cp $LIB /tmp/orig.so
for rep in 1 2 3; do
  for v in A B; do
    f=$A; [ $v = B ] && f=$B
    cp $f $LIB
    python bench.py --no_cpu_baseline --no_jta --no_policy --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'env_only', d['env_step_only']['value'])"
  done
done
cp /tmp/orig.so $LIB
