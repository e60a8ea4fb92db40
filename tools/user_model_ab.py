"""Generic kernels vs the kernel set a user model gets through the DEFAULT Engine(...) path (round 6: the cached set if there is one,
otherwise the generic kernels now and a background compile whose result the NEXT Engine of the model picks up):
    python tools/user_model_ab.py tests/golden/user_tree.npz [n_envs] [substeps]
operator-level launches (dsim_step_forward / backward), minimum of 5 rounds of 20."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffrl_amd.engine import Engine
from diffrl_amd.template import ArticulationTemplate
from test_edge_cases_cpu import _tree_states

path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
S = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
t = ArticulationTemplate.load(path)
q, qd, act = _tree_states(t, np.random.default_rng(3), n)
T = lambda a: torch.tensor(a, device=dev).reshape(-1)
q, qd, act = T(q), T(qd), T(act)
gq, gqd = torch.randn_like(q), torch.randn_like(qd)
from diffrl_amd import specialise as _sp
os.environ.pop("DSIM_AUTO_SPECIALISE", None)
for auto in (False, "default (first Engine)", "default (next Engine)"):
    t0 = time.time()
    eng = Engine(t, dev, specialise=False) if auto is False else Engine(t, dev)
    setup = time.time() - t0
    if auto == "default (first Engine)":
        t1 = time.time()
        _sp.wait(t)   # (a background compile in flight, if the cache had no set for this model)
        print("   background compile waited for: %.1f s" % (time.time() - t1))
    best = [1e9, 1e9]
    for r in range(5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        qo, qdo, ck = eng.forward(q, qd, act, None, 1 / 60.0, S, S, True)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            qo, qdo, ck = eng.forward(q, qd, act, None, 1 / 60.0, S, S, True)
        ev[1].record()
        for _ in range(20):
            g = eng.backward(ck, act, None, 1 / 60.0, S, S, gq, gqd)
        ev[2].record()
        torch.cuda.synchronize()
        best = [min(best[0], ev[0].elapsed_time(ev[1]) / 20), min(best[1], ev[1].elapsed_time(ev[2]) / 20)]
    print("%-34s %s N=%d substeps=%d: fwd %.4f ms  bwd %.4f ms -> %.3f M steps/s   (Engine() %.1f s)"
          % ("%s: variant %d" % (auto, eng.variant) if auto else "generic kernels (specialise=False)", os.path.basename(path), n, S,
             best[0], best[1], n / (best[0] + best[1]) / 1e3, setup))
