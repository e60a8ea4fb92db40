#!/bin/bash
# Developer tool (runs on the GPU box): tools/ab.sh <env> <N> lib1.so lib2.so ... -> kernel timings + output hashes per library
ENVN=$1; N=$2; shift 2
for lib in "$@"; do
  echo "== $lib"
  DSIM_LIB=$PWD/$lib python tools/gpu_quick.py $ENVN $N 2>&1 | grep -v amdgpu.ids | grep -v "^AMD"
  DSIM_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --env $ENVN --envs-per-gpu ${N%%,*} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   bench %.4g env-steps/s  fwd %.4f ms  bwd %.4f ms' % (d['value'], d['roofline']['fwd_kernel_ms'], d['roofline']['kernel_ms']))
"
done
