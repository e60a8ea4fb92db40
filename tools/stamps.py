"""Developer tool (GPU box): cycle stamps of workgroup 0's main wave at the phase boundaries of the PRODUCT executor
(helper wave included), from a library built with -DDSIM_STAMPS (tools/dev_build.sh stamps all -DDSIM_STAMPS).
usage: DSIM_LIB=tools/libdsim_stamps.so python tools/stamps.py <env> [N]   ->  per-tag cycle means of one forward and one
adjoint env-step launch.  clock64 drains the LDS queue at every stamp: phases look a little longer than they are."""
import collections
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from diffrl_amd import capi
from emu_lib import env_spec_for
from oracle_lib import golden, template_from_golden

L = capi.lib()
env = sys.argv[1] if len(sys.argv) > 1 else "ant"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
t = template_from_golden(env); g = golden(env + "_step")
S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
dev = torch.device("cuda:0")
desc, keep = capi.make_desc(t)
h = C.c_void_p()
capi.check(L.dsim_model_create(C.byref(desc), C.byref(h)))
spec, sc = env_spec_for(env, t)
sc_dev = torch.tensor(sc, device=dev)
spec.act_scale = sc_dev.data_ptr()
reps = N // g["q_in"].shape[0] + 1
q = torch.tensor(np.tile(g["q_in"], (reps, 1))[:N], device=dev).reshape(-1)
qd = torch.tensor(np.tile(g["qd_in"], (reps, 1))[:N], device=dev).reshape(-1)
a = torch.zeros((N, spec.n_act), device=dev)
qo, qdo = torch.empty_like(q), torch.empty_like(qd)
obs, rew = torch.empty((N, spec.n_obs), device=dev), torch.empty(N, device=dev)
ck = torch.empty((N, int(L.dsim_ckpt_floats_mm(h, S, mm))), device=dev)
gq, gqd, go, gr = torch.randn_like(q), torch.randn_like(qd), torch.randn_like(obs), torch.randn_like(rew)
gqi, gqdi, ga = torch.empty_like(q), torch.empty_like(qd), torch.empty_like(a)
p = lambda x: C.c_void_p(x.data_ptr())
CAP = 16384
names = {0: "prologue", 1: "fwd_kin", 2: "fwd_ext", 3: "fwd_dyn/tau", 4: "fwd_mass", 5: "fwd_solve", 6: "fwd_integ", 7: "bwd_joint",
         9: "bwd_mass", 10: "bwd_bodies"}
L.dsim_debug_stamps.argtypes = [C.c_void_p, C.c_int]
for backward in (0, 1):
    for it in range(3):
        if it == 2:   # (read-and-clear: the measured launch starts from an empty buffer)
            capi.check(L.dsim_debug_stamps(np.zeros(2 * CAP, np.int64).ctypes.data_as(C.c_void_p), 2 * CAP))
        if backward:
            capi.check(L.dsim_env_step_backward(h, C.byref(spec), N, p(ck), p(a), C.c_float(dt), S, mm, p(gq), p(gqd), p(go), p(gr),
                                                None, p(gqi), p(gqdi), p(ga), None))
        else:
            capi.check(L.dsim_env_step_forward(h, C.byref(spec), N, p(q), p(qd), p(a), C.c_float(dt), S, mm, p(qo), p(qdo), p(obs),
                                               p(rew), p(ck), None, None))
        torch.cuda.synchronize()
    buf = np.zeros(2 * CAP, np.int64)
    capi.check(L.dsim_debug_stamps(buf.ctypes.data_as(C.c_void_p), 2 * CAP))
    for wave, lo in (("main wave", 0), ("helper wave", 8192)):
        clk, tag = buf[lo:lo + 8192], buf[CAP + lo:CAP + lo + 8192]
        n = int((clk != 0).sum())
        if n < 2:
            continue
        d = np.diff(clk[:n])
        tags = tag[1:n]
        print("==", env, "N", N, "adjoint" if backward else "forward", wave, "stamps", n, "total cycles (stamp clock)", int(clk[n - 1] - clk[0]))
        agg = collections.OrderedDict()
        for tg, x in zip(tags.tolist(), d.tolist()):
            agg.setdefault(tg, []).append(x)
        tot = collections.Counter()
        for tg, xs in agg.items():
            tot[names.get(tg // 100, "?")] += sum(xs)
            print("  %-12s stamp %2d: n=%3d mean %7.0f  min %6d max %6d" % (names.get(tg // 100, "?"), tg % 100, len(xs), sum(xs) / len(xs), min(xs), max(xs)))
        print("  totals:", dict(tot))
