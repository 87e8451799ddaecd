#!/bin/bash
# Build variants/attn_NAME.so with only the attention unit (extra hipcc flags in $2...).  Usage: bash tools/exp/mk_attn_variant.sh NAME [-DFOO=1 ...]
N=$1; shift
mkdir -p variants /tmp/gv
hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I emloco_amd/csrc -c emloco_amd/csrc/attention_capi.hip -o /tmp/gv/attn_$N.o 2>&1 | grep -E "error|spill"
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/attn_$N.so /tmp/gv/attn_$N.o && echo built variants/attn_$N.so
