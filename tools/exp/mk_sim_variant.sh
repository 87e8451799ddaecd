#!/bin/bash
# Build variants/simv_NAME.so: the whole library with the rigid-body unit recompiled with extra hipcc flags ($2...), the other units' objects
# reused from emloco_amd/lib.  Use: EMLOCO_LIB=$PWD/variants/simv_NAME.so python ... (tools/exp/ab_lib_env.sh)
N=$1; shift
mkdir -p variants /tmp/gv
hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-function -I emloco_amd/csrc -c emloco_amd/csrc/sim_capi.hip -o /tmp/gv/simv_$N.o 2>&1 | grep -E "error|spill"
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/simv_$N.so /tmp/gv/simv_$N.o emloco_amd/lib/task_capi.o emloco_amd/lib/predictor_capi.o emloco_amd/lib/attention_capi.o emloco_amd/lib/ffn_capi.o emloco_amd/lib/ppo_capi.o && echo built variants/simv_$N.so
