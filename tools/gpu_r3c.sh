#!/bin/bash
# round 3, GPU call C: Humanoid after trunk-in-registers / bpermute rounds / unrolled checkpoint copy; full GPU test suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "### quick timings, production library"; python tools/gpu_quick.py 2>&1 | grep -v amdgpu.ids
echo "### stamps humanoid"; DSIM_LIB=$PWD/tools/libdsim_stamps.so python tools/stamps.py humanoid 1024 2>&1 | grep -v amdgpu.ids
echo "### gpu tests"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
} > gpurun_out/r3c.log 2>&1
tail -30 gpurun_out/r3c.log
