#!/bin/bash
# GPU box: the reference's unmodified callers (scratch copy build/ref_callers, tools/setup_ref_callers.sh) against this
# repository's envs / dflex through dropin/: BPTT (algorithms/bptt.py), PPO through rl_games (examples/train_rl.py), SHAC.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
# (run from a writable copy: rl_games creates ./train_dir in the working directory, and the snapshot may be read-only)
rm -rf /tmp/ref_callers && cp -r $ROOT/build/ref_callers /tmp/ref_callers && chmod -R u+w /tmp/ref_callers
cd /tmp/ref_callers/examples
export PYTHONPATH=$ROOT/dropin:/tmp/ref_callers/externals/rl_games
mkdir -p $ROOT/gpurun_out
run() { name=$1; shift; echo "### $name: $*"; ( time timeout 900 "$@" ) > $ROOT/gpurun_out/callers_$name.log 2>&1; echo "rc=$?"; tail -4 $ROOT/gpurun_out/callers_$name.log; }
[ -z "$ONLY_PPO" ] && run bptt_ant python train_bptt.py --cfg ./cfg/bptt/ant_short.yaml --logdir /tmp/logs/bptt_ant --no-time-stamp
run ppo_ant python train_rl.py --cfg ./cfg/ppo/ant_short.yaml --logdir /tmp/logs/ppo_ant --no-time-stamp
[ -z "$ONLY_PPO" ] && run shac_ant python train_shac.py --cfg ./cfg/shac/ant_short.yaml --logdir /tmp/logs/shac_ant --no-time-stamp
# where the wall time of the unmodified SHAC goes at its shipped size (64 environments): cumulative host profile
if [ -z "$ONLY_PPO" ]; then
  echo "### shac_ant host profile"
  timeout 900 python -m cProfile -o /tmp/shac.prof train_shac.py --cfg ./cfg/shac/ant_short.yaml --logdir /tmp/logs/shac_prof --no-time-stamp > /dev/null 2>&1
  python -c "
import pstats
s = pstats.Stats('/tmp/shac.prof'); s.sort_stats('cumulative').print_stats(45)" 2>&1 | cut -c1-180 > $ROOT/gpurun_out/callers_shac_hostprofile.log
  tail -5 $ROOT/gpurun_out/callers_shac_hostprofile.log
fi
