"""Developer tool (GPU box): two-environments-per-wave kernels (DSIM_PAIR=1) against the one-environment kernels, same inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle_lib import golden, template_from_golden
from diffrl_amd.engine import Engine

dev = torch.device("cuda:0")
for env in sys.argv[1].split(","):
    for N in (1, 7, 64):
        t = template_from_golden(env); g = golden(env + "_step")
        S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
        reps = N // g["q_in"].shape[0] + 1
        q = torch.tensor(np.tile(g["q_in"], (reps, 1))[:N], device=dev).reshape(-1)
        qd = torch.tensor(np.tile(g["qd_in"], (reps, 1))[:N], device=dev).reshape(-1)
        a = torch.tensor(np.tile(g["act_in"], (reps, 1))[:N], device=dev).reshape(-1)
        torch.manual_seed(0)
        gq, gqd = torch.randn_like(q), torch.randn_like(qd)
        outs = []
        for pair in ("0", "1"):
            os.environ["DSIM_PAIR"] = pair; os.environ["DSIM_HELPER"] = "0"
            eng = Engine(t, dev)
            qo, qdo, ck = eng.forward(q, qd, a, None, dt, S, mm, True)
            b = eng.backward(ck, a, None, dt, S, mm, gq, gqd)
            torch.cuda.synchronize()
            outs.append([x.detach().cpu().numpy() for x in (qo, qdo, ck) + tuple(y for y in b if y is not None)])
        names = ["q", "qd", "ckpt", "gq", "gqd", "gact", "gm"]
        msg = []
        for nm, x, y in zip(names, outs[0], outs[1]):
            d = np.abs(x - y).max(); r = d / (np.abs(x).max() + 1e-30)
            msg.append("%s %.2e/%.2e%s" % (nm, d, r, "" if np.isfinite(y).all() else " NONFINITE"))
        print(env, "N=%d" % N, "  ".join(msg))
