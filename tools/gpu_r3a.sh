#!/bin/bash
# round 3, GPU call A: log-depth kinematics A/B (old library vs scan builds) + parity of the scan kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OLD=diffrl_amd/csrc/libdsim_hip.so
{
echo "### AB humanoid"; tools/ab.sh humanoid 1024 $OLD tools/libdsim_scan5.so
echo "### humanoid scan, no helper"; DSIM_HELPER=0 tools/ab.sh humanoid 1024 tools/libdsim_scan5.so
echo "### AB ant"; tools/ab.sh ant 1024 $OLD tools/libdsim_scan5.so tools/libdsim_scan3.so
echo "### AB hopper"; tools/ab.sh hopper 1024 $OLD tools/libdsim_scan5.so
echo "### AB cheetah"; tools/ab.sh cheetah 1024 $OLD tools/libdsim_scan5.so
echo "### parity tests with the scan library"
DSIM_LIB=$PWD/tools/libdsim_scan5.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_envs.py -x -q -k "humanoid or hopper or cheetah or ant" 2>&1 | tail -15
} > gpurun_out/r3a.log 2>&1
tail -60 gpurun_out/r3a.log
