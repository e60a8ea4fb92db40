"""Developer tool: full vs lean checkpoint mode, kernels only (forward with checkpoint + adjoint), several N."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle_lib import golden, template_from_golden
from diffrl_amd.engine import Engine
dev = torch.device("cuda:0")
for env, Ns in (("ant", (1024, 8192)), ("humanoid", (1024, 4096)), ("snu", (512,))):
    t = template_from_golden(env); g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    for N in Ns:
        reps = N // g["q_in"].shape[0] + 1
        q = torch.tensor(np.tile(g["q_in"], (reps, 1))[:N], device=dev).reshape(-1)
        qd = torch.tensor(np.tile(g["qd_in"], (reps, 1))[:N], device=dev).reshape(-1)
        a = torch.tensor(np.tile(g["act_in"], (reps, 1))[:N], device=dev).reshape(-1)
        m = torch.tensor(np.tile(g["muscle_act_in"], (reps, 1))[:N], device=dev).reshape(-1) if "muscle_act_in" in g else None
        gq, gqd = torch.randn_like(q), torch.randn_like(qd)
        for mode in ("full", "lean"):
            eng = Engine(t, dev, ckpt_mode=mode)
            for _ in range(3):
                qo, qdo, ck = eng.forward(q, qd, a, m, dt, S, mm, True); b = eng.backward(ck, a, m, dt, S, mm, gq, gqd)
            torch.cuda.synchronize()
            e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            K = 20
            e0.record()
            for _ in range(K): qo, qdo, ck = eng.forward(q, qd, a, m, dt, S, mm, True)
            e1.record()
            for _ in range(K): b = eng.backward(ck, a, m, dt, S, mm, gq, gqd)
            e2.record(); torch.cuda.synchronize()
            tf, tb = e0.elapsed_time(e1) / K, e1.elapsed_time(e2) / K
            print("%-9s N=%5d %-4s ckpt %7.1f KB/env-step  fwd %.3f ms  bwd %.3f ms  -> %.3e env-steps/s" % (env, N, mode, ck.shape[1] * 4 / 1024, tf, tb, N / ((tf + tb) * 1e-3)))
