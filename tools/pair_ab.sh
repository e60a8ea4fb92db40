#!/bin/bash
# Developer tool (GPU box): two environments per wavefront (DSIM_MODE_PAIR, the default beyond the helper-wave capacity) against
# one (DSIM_PAIR=0), same library: bit-identity check, operator-kernel timings and the bench rollout at 8192 environments.
# usage: tools/pair_ab.sh [library.so]      (default: the in-tree library)
[ -n "$1" ] && export DSIM_LIB=$PWD/$1
echo "== pair_check"; timeout 300 python tools/pair_check.py ant,hopper,cheetah,cartpole 2>&1 | grep -v amdgpu.ids
for p in "" 0; do echo "== gpu_quick ant DSIM_PAIR=${p:-auto}"; env ${p:+DSIM_PAIR=$p} timeout 300 python tools/gpu_quick.py ant 1536,2048,4096,8192,16384 2>&1 | grep "^ant"; done
for e in hopper cheetah cartpole; do for p in "" 0; do env ${p:+DSIM_PAIR=$p} timeout 300 python tools/gpu_quick.py $e 8192 2>&1 | grep "N=" | sed "s/^/PAIR=${p:-auto} /"; done; done
for p in "" 0; do echo "== bench 8192 DSIM_PAIR=${p:-auto}"; env ${p:+DSIM_PAIR=$p} timeout 300 python bench.py --envs-per-gpu 8192 --no-cpu-baseline --no-other-configs --no-extras --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
