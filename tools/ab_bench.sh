#!/bin/bash
# Developer tool (GPU box): the bench rollout (env kernels, graph replay) with several libraries in alternation.
# usage: tools/ab_bench.sh <env> <envs> <rounds> lib1.so lib2.so ...
ENVN=$1; N=$2; R=$3; shift 3
for r in $(seq $R); do for l in "$@"; do
  DSIM_LIB=$PWD/$l python bench.py --env $ENVN --envs-per-gpu $N --no-cpu-baseline --no-other-configs --no-extras --steps 10 --warmup 3 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$l $ENVN $N  %.4g env-steps/s  ms/step min %.4f  fwd %.4f ms  bwd %.4f ms' % (d['value'], d.get('ms_per_step_min', 0), r['fwd_kernel_ms'], r['kernel_ms']))"
done; done
