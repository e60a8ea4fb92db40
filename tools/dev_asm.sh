#!/bin/bash
# Developer tool: device assembly of a subset build.  tools/dev_asm.sh <out.s> [models] [extra flags]
OUT=$1; shift
MODELS=${1:-Ant}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
X=""; for m in ${MODELS//,/ }; do X="$X X($m)"; done
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
  "-DDSIM_STATIC_VARIANTS(X)=$X" "$@" --offload-device-only -S "$ROOT/diffrl_amd/csrc/dsim_hip.hip" -o "$OUT"
