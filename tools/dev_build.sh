#!/bin/bash
# Developer tool: A/B library builds.  tools/dev_build.sh <tag> [models|all] [extra hipcc flags...]
#   -> tools/libdsim_<tag>.so with only the named specialised variants (default: Ant; generic kernels always);
#   run with DSIM_LIB=tools/libdsim_<tag>.so.
TAG=$1; shift
MODELS=${1:-Ant}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VARIANTS=""
if [ "$MODELS" != "all" ]; then
  X=""; for m in ${MODELS//,/ }; do X="$X X($m)"; done
  VARIANTS="-DDSIM_STATIC_VARIANTS(X)=$X"
fi
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -fno-slp-vectorize \
  -mllvm -amdgpu-sched-strategy=max-ilp ${VARIANTS:+"$VARIANTS"} "$@" \
  "$ROOT/diffrl_amd/csrc/dsim_hip.hip" -o "$ROOT/tools/libdsim_$TAG.so"
