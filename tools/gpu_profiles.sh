#!/bin/bash
# GPU box: kernel trace + counter passes of the single-GPU configurations at the current kernel sources
#   tools/gpu_profiles.sh <tag> [configs...]   configs: ant humanoid snu ant8192 antmm1 (default: all five)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}; shift
CFGS=${@:-ant humanoid snu ant8192 antmm1}
for c in $CFGS; do
  case $c in
    ant) bash tools/profile.sh $TAG ant 1024 > /dev/null 2>&1;;
    humanoid) bash tools/profile.sh $TAG humanoid 1024 > /dev/null 2>&1;;
    snu) bash tools/profile.sh $TAG snu 512 > /dev/null 2>&1;;
    ant8192) bash tools/profile.sh $TAG ant 8192 > /dev/null 2>&1;;
    antmm1) bash tools/profile.sh $TAG ant 1024 1 > /dev/null 2>&1;;
  esac
done
for d in gpurun_out/prof_${TAG}_*; do echo "== $d"; grep -E "dsim_env_(fwd|bwd)|nan_to" $d/summary.txt | cut -c1-200 | head -8; done
