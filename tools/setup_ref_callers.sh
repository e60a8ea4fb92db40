#!/bin/bash
# Copies the reference's CALLERS (algorithms / models / utils / optim / examples / externals/rl_games), byte for byte, into the
# git-ignored scratch directory build/ref_callers so that they can be run UNMODIFIED on the GPU box (where /root/reference does
# not exist) against this repository's envs / dflex through dropin/.  Only shortened copies of the yaml configs are added
# (<env>_short.yaml: fewer epochs).  Nothing under build/ is part of the repository.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${1:-/root/reference}
DST=$ROOT/build/ref_callers
rm -rf "$DST"; mkdir -p "$DST/externals"
for d in algorithms models utils optim examples; do [ -d "$REF/$d" ] && cp -r "$REF/$d" "$DST/$d"; done
cp -r "$REF/externals/rl_games" "$DST/externals/rl_games"
python - "$DST" <<'PY'
import sys, yaml, os
dst = sys.argv[1]
def short(algo, env, **over):
    p = os.path.join(dst, "examples", "cfg", algo, env + ".yaml")
    cfg = yaml.safe_load(open(p))
    c = cfg["params"]["config"]
    for k, v in over.items():
        c[k] = v
    yaml.safe_dump(cfg, open(p.replace(".yaml", "_short.yaml"), "w"))
short("bptt", "ant", max_epochs=20, steps_num=128)       # shipped: 2000 epochs x 1000-step rollouts
short("bptt", "cartpole_swing_up", max_epochs=20)
short("ppo", "ant", max_epochs=30, save_frequency=1000)
short("shac", "ant", max_epochs=40, save_interval=1000)
PY
echo "reference callers copied to $DST"
