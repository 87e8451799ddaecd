#!/bin/bash
# the round's whole evidence set on one box: bash tools/exp/collect_all.sh r04   (through gpurun; summaries land in gpurun_out/<round>/final/)
R=${1:-r04}
F=gpurun_out/$R/final
mkdir -p $F
python bench.py > $F/bench_default.log 2>$F/bench_default.err
bash tools/collect_profiles.sh $R > $F/collect.log 2>&1
bash tools/exp/prof_env2.sh $F/env_step_trace.txt > /dev/null 2>&1
bash tools/exp/prof_env_disc.sh $F/env_step_trace_disc.txt > /dev/null 2>&1
bash tools/exp/prof_jta.sh $F/jta_step_kernels.txt > /dev/null 2>&1
JTA_PRECISION=bf16 bash tools/exp/prof_jta.sh $F/jta_bf16_step_kernels.txt > /dev/null 2>&1
bash tools/exp/prof_ppo.sh $F/ppo_step_kernels.txt 1 > /dev/null 2>&1
python tools/exp/chain_prof.py > $F/chain_prof.txt 2>&1
bash tools/exp/sq_diag.sh $F/sim_step_sq_counters.txt > /dev/null 2>&1
tail -c 600 $F/bench_default.log
