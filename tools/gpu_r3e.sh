#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "### quick timings, production library"; python tools/gpu_quick.py 2>&1 | grep -v amdgpu.ids
for e in ant humanoid; do echo "### stamps $e"; DSIM_LIB=$PWD/tools/libdsim_stamps.so python tools/stamps.py $e 1024 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r3e.log 2>&1
bash tools/gpu_tests.sh
