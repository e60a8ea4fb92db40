#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for lib in tools/libdsim_prev.so diffrl_amd/csrc/libdsim_hip.so; do
  echo "== $lib"; DSIM_LIB=$PWD/$lib python tools/gpu_quick.py ant 1024,2048,8192 2>&1 | grep -v amdgpu.ids | grep -v "^AMD"
  DSIM_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-extras 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   bench %.4g env-steps/s  fwd %.4f ms  bwd %.4f ms' % (d['value'], d['roofline']['fwd_kernel_ms'], d['roofline']['kernel_ms']))
"
done
} > gpurun_out/ab_ant.log 2>&1
cat gpurun_out/ab_ant.log
