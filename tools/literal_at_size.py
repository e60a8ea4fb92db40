import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from oracle_lib import golden, template_from_golden
from diffrl_amd.engine import Engine
dev = torch.device("cuda:0")
for env, N in (("ant", 1024), ("humanoid", 1024), ("snu", 512), ("ant", 8192)):
    t = template_from_golden(env); g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    reps = N // g["q_in"].shape[0] + 1
    T = lambda a: torch.tensor(np.tile(a, (reps, 1))[:N], device=dev).reshape(-1)
    q, qd, a = T(g["q_in"]), T(g["qd_in"]), T(g["act_in"])
    m = T(g["muscle_act_in"]) if "muscle_act_in" in g else None
    gq, gqd = T(g["gq_out"]), T(g["gqd_out"])
    eng = Engine(t, dev)
    qo, qdo, ck = eng.forward(q, qd, a, m, dt, S, mm, True)
    for lit in (False, True):
        for _ in range(2): r = eng.backward(ck, a, m, dt, S, mm, gq, gqd, literal=lit)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): r = eng.backward(ck, a, m, dt, S, mm, gq, gqd, literal=lit)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10
        n0 = g["q_in"].shape[0]
        err = np.abs(r[0].view(N, -1)[:n0].cpu().numpy() - g["gq_in"]).max() / np.abs(g["gq_in"]).max()
        rep_ok = torch.equal(r[0].view(N, -1)[:n0], r[0].view(N, -1)[n0 * (reps - 2):n0 * (reps - 1)]) if reps > 2 else True
        print("%-8s N=%5d literal=%-5s backward %.3f ms  unprojected |gq - ref| %.2e  replicas equal %s" % (env, N, lit, el * 1e3, err, rep_ok))
