# A/B of two builds of the library on the rollout legs (headline, env.step alone): bash tools/exp/ab_lib_env.sh libA.so libB.so
for rep in 1 2 3; do for L in "$@"; do EMLOCO_LIB=$PWD/$L python bench.py --steps 400 --warmup 40 --no_cpu_baseline --no_ppo --no_jta --no_policy 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$L', 'headline', round(d['value']/1e6,3), 'M', d['ms_per_step'], 'ms; env_step_only', round(d.get('env_step_only',{}).get('value',0)/1e6,3) if isinstance(d.get('env_step_only'),dict) else d.get('env_step_only'))"; done; done
