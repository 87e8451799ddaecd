mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_default.log 2>gpurun_out/final/bench_default.err
bash tools/collect_profiles.sh r03 > gpurun_out/final/collect.log 2>&1
bash tools/exp/prof_env2.sh gpurun_out/final/env_step_trace.txt > /dev/null 2>&1
bash tools/exp/prof_jta.sh gpurun_out/final/jta_step_kernels.txt > /dev/null 2>&1
JTA_PRECISION=bf16 bash tools/exp/prof_jta.sh gpurun_out/final/jta_bf16_step_kernels.txt > /dev/null 2>&1
python tools/exp/chain_prof.py > gpurun_out/final/chain_prof.txt 2>&1
bash tools/exp/sq_diag.sh gpurun_out/final/sim_step_sq_counters.txt > /dev/null 2>&1
tail -c 600 gpurun_out/final/bench_default.log
