#!/bin/bash
# A/B: cross-stream events of the rollout chain with a system-scope release (torch.cuda.Event) or a device-scope release
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/ab_event_scope.txt
: > $OUT
for rep in 1 2; do
for v in system device; do
  echo "== EMLOCO_EVENT_SCOPE=$v" >> $OUT
  EMLOCO_EVENT_SCOPE=$v timeout 300 python bench.py --no_jta --no_policy --no_ppo 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('env_step_only'))" >> $OUT
done; done
cat $OUT
