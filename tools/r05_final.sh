#!/bin/bash
mkdir -p gpurun_out
{
echo "# SNUHumanoid 512 environments, operator-level kernels (tools/ab_min.py, minimum of 3 alternating rounds), MI355X, round 5"
echo "# libraries: the shipped kernel set with ONE round-5 change taken out each (tools/dev_build.sh <tag> Snu <flag>)"
python tools/ab_min.py snu 512 3 tools/libdsim_snu_final.so tools/libdsim_snu_abl_nooverlap.so tools/libdsim_snu_abl_classicgj.so tools/libdsim_snu_abl_exact.so tools/libdsim_snu_abl_r4.so
} > gpurun_out/r05_ablation_snu.txt 2>&1
DSIM_LIB=$PWD/tools/libdsim_snu_stamps.so python tools/stamps.py snu 512 > gpurun_out/r05_stamps_snu.txt 2>&1
DSIM_HELPER=1 DSIM_LIB=$PWD/tools/libdsim_ant_stamps.so python tools/stamps.py ant 1024 > gpurun_out/r05_stamps_ant.txt 2>&1
DSIM_HELPER=1 DSIM_LIB=$PWD/tools/libdsim_hum_stamps.so python tools/stamps.py humanoid 1024 > gpurun_out/r05_stamps_humanoid.txt 2>&1
cat gpurun_out/r05_ablation_snu.txt
{ echo "# python tools/snu_refresh_probe.py, MI355X, shipped kernels, minimum of 3 x 20 launches"; python tools/snu_refresh_probe.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05_snu_refresh_probe.txt
bash tools/gpu_profiles.sh r05f ant humanoid snu ant8192 antmm1
