#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for lib in diffrl_amd/csrc/libdsim_hip.so tools/libdsim_snu1.so tools/libdsim_snu2.so; do
  echo "== $lib"; DSIM_LIB=$PWD/$lib python tools/gpu_quick.py snu 512,1024,2048 2>&1 | grep -v amdgpu.ids | grep -v "^AMD"
done
} > gpurun_out/snu_waves.log 2>&1
cat gpurun_out/snu_waves.log
