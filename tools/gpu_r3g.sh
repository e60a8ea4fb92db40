#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/gpu_quick.py 2>&1 | grep -v amdgpu.ids > gpurun_out/quick.log; cat gpurun_out/quick.log
bash tools/gpu_tests.sh
bash tools/gpu_callers.sh
