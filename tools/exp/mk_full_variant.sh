#!/bin/bash
# Build variants/full_NAME.so: the whole library with the predictor unit recompiled with extra hipcc flags ($2...), the other units' objects
# reused from emloco_amd/lib (run `python -m emloco_amd.build` first).  Use: EMLOCO_LIB=$PWD/variants/full_NAME.so python ...
N=$1; shift
mkdir -p variants /tmp/gv
hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-function -I emloco_amd/csrc -c emloco_amd/csrc/predictor_capi.hip -o /tmp/gv/full_$N.o 2>&1 | grep -E "error|spill"
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/full_$N.so emloco_amd/lib/sim_capi.o emloco_amd/lib/task_capi.o emloco_amd/lib/attention_capi.o emloco_amd/lib/ffn_capi.o emloco_amd/lib/ppo_capi.o /tmp/gv/full_$N.o && echo built variants/full_$N.so
