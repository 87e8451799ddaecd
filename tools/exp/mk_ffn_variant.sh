#!/bin/bash
# Build variants/NAME.so with only the chained feed-forward unit recompiled (extra hipcc flags in $2...), the other objects reused.
# Usage: bash tools/exp/mk_ffn_variant.sh NAME [-DFOO=1 ...]
N=$1; shift
mkdir -p variants /tmp/gv
hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I emloco_amd/csrc -c emloco_amd/csrc/ffn_capi.hip -o /tmp/gv/ffn_$N.o 2>&1 | grep -E "error|spill"
hipcc --offload-arch=gfx950 -shared -fPIC -o variants/ffn_$N.so /tmp/gv/ffn_$N.o && echo built variants/ffn_$N.so
