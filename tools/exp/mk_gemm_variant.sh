#!/bin/bash
# Build variants/NAME.so with only the predictor unit recompiled (extra hipcc flags in $2...), the other objects reused.
# Usage: bash tools/exp/mk_gemm_variant.sh NAME [-DFOO=1 ...]
N=$1; shift
mkdir -p variants /tmp/gv
hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I emloco_amd/csrc -c emloco_amd/csrc/predictor_capi.hip -o /tmp/gv/$N.o 2>&1 | grep -E "error|spill" 
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$N.so emloco_amd/lib/sim_capi.o emloco_amd/lib/task_capi.o /tmp/gv/$N.o && echo built variants/$N.so
