#!/bin/bash
# Build a named variant of libemloco_hip.so (+ its -DEMLOCO_SIM_PROFILE=1 twin) here, into variants/ (travels with gpurun).
# Usage: bash tools/exp/mk_variant.sh NAME [extra hipcc flags]
N=$1; shift
mkdir -p variants
EMLOCO_HIPCC_EXTRA="-DEMLOCO_SIM_PROFILE=1 $*" python -m emloco_amd.build --force > /dev/null 2>&1 || { echo "profile build failed"; exit 1; }
cp emloco_amd/lib/libemloco_hip.so variants/${N}_prof.so
EMLOCO_HIPCC_EXTRA="$*" python -m emloco_amd.build --force > /dev/null 2>&1 || { echo "build failed"; exit 1; }
cp emloco_amd/lib/libemloco_hip.so variants/${N}.so
echo built variants/$N
