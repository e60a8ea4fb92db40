#!/bin/bash
# GPU box: kernel trace + counter passes of the three single-GPU configurations at the current kernel sources
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
bash tools/profile.sh $TAG ant 1024 > /dev/null 2>&1
bash tools/profile.sh $TAG humanoid 1024 > /dev/null 2>&1
bash tools/profile.sh $TAG snu 512 > /dev/null 2>&1
for e in ant humanoid snu; do echo "== $e"; grep -E "dsim_env_(fwd|bwd)" gpurun_out/prof_${TAG}_$e/summary.txt | cut -c1-260 | head -12; done
