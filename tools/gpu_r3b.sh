#!/bin/bash
# round 3, GPU call B: cycle stamps of the product executor (main wave of workgroup 0), forward + adjoint
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for e in humanoid ant snu; do
  N=1024; [ $e = snu ] && N=512
  echo "### stamps $e"; DSIM_LIB=$PWD/tools/libdsim_stamps.so python tools/stamps.py $e $N 2>&1 | grep -v amdgpu.ids
done
echo "### stamps humanoid, no helper"; DSIM_HELPER=0 DSIM_LIB=$PWD/tools/libdsim_stamps.so python tools/stamps.py humanoid 1024 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3b.log 2>&1
tail -5 gpurun_out/r3b.log
