export DSIM_LIB=$PWD/tools/libdsim_p2.so
echo "== pair_check (p2)"; timeout 300 python tools/pair_check.py ant,hopper,cheetah,cartpole 2>&1 | grep -v amdgpu.ids
for p in "" 0; do echo "== gpu_quick p2 DSIM_PAIR=$p"; env ${p:+DSIM_PAIR=$p} timeout 300 python tools/gpu_quick.py ant 1536,2048,4096,8192,16384 2>&1 | grep "^ant"; done
echo "== gpu_quick hopper/cheetah 8192"; for e in hopper cheetah cartpole; do for p in "" 0; do env ${p:+DSIM_PAIR=$p} timeout 300 python tools/gpu_quick.py $e 8192 2>&1 | grep "N=" | sed "s/^/PAIR=$p /"; done; done
for l in p2 p3; do export DSIM_LIB=$PWD/tools/libdsim_$l.so; for p in "" 0; do echo "== bench 8192 lib $l DSIM_PAIR=$p"; env ${p:+DSIM_PAIR=$p} timeout 300 python bench.py --envs-per-gpu 8192 --no-cpu-baseline --no-other-configs --no-extras --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
