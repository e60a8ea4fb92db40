"""Developer tool: per-phase cycle counts of workgroup 0 for the fused fwd / bwd kernels (clock64 stamps after
every phase).  Builds a separate library with -DDSIM_ENABLE_PHASE_TIMER; not part of the product."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from diffrl_amd import capi
from emu_lib import env_spec_for
from oracle_lib import golden, template_from_golden

so = os.path.join(ROOT, "tools", "libdsim_timer.so")
if not os.environ.get("DSIM_TIMER_PREBUILT"):   # (the GPU box has no time to spare for a 2-minute compile: build it beforehand with tools/dev_build.sh timer all -DDSIM_ENABLE_PHASE_TIMER and copy it)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                           "-DDSIM_ENABLE_PHASE_TIMER", os.path.join(ROOT, "diffrl_amd", "csrc", "dsim_hip.hip"), "-o", so])
L = C.CDLL(so)
L.dsim_last_error.restype = C.c_char_p
env = sys.argv[1] if len(sys.argv) > 1 else "ant"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
t = template_from_golden(env); g = golden(env + "_step")
S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
dev = torch.device("cuda:0")
desc, keep = capi.make_desc(t)
h = C.c_void_p()
assert L.dsim_model_create(C.byref(desc), C.byref(h)) == 0, L.dsim_last_error()
spec, sc = env_spec_for(env, t)
sc_dev = torch.tensor(sc, device=dev)
spec.act_scale = sc_dev.data_ptr()
reps = N // g["q_in"].shape[0] + 1
q = torch.tensor(np.tile(g["q_in"], (reps, 1))[:N], device=dev).reshape(-1)
qd = torch.tensor(np.tile(g["qd_in"], (reps, 1))[:N], device=dev).reshape(-1)
a = torch.zeros((N, spec.n_act), device=dev)
qo, qdo = torch.empty_like(q), torch.empty_like(qd)
obs, rew = torch.empty((N, spec.n_obs), device=dev), torch.empty(N, device=dev)
L.dsim_ckpt_floats_mm.restype = C.c_int64
L.dsim_ckpt_floats_mm.argtypes = [C.c_void_p, C.c_int, C.c_int]
ck = torch.empty((N, int(L.dsim_ckpt_floats_mm(h, S, mm))), device=dev)
gq, gqd, go, gr = torch.randn_like(q), torch.randn_like(qd), torch.randn_like(obs), torch.randn_like(rew)
gqi, gqdi, ga = torch.empty_like(q), torch.empty_like(qd), torch.empty_like(a)
cap = 4096
p = lambda x: C.c_void_p(x.data_ptr())
for backward in (0, 1):
    stamps = torch.zeros(2 * cap, dtype=torch.int64, device=dev)
    for _ in range(2):
        rc = L.dsim_debug_phase_timer(h, C.byref(spec), N, backward, p(q), p(qd), p(a), C.c_float(dt), S, mm, p(qo), p(qdo),
                                      p(obs), p(rew), p(ck), p(gq), p(gqd), p(go), p(gr), p(gqi), p(gqdi), p(ga), p(stamps),
                                      cap, None)
        assert rc == 0, L.dsim_last_error()
        torch.cuda.synchronize()
    st = stamps.cpu().numpy()
    n = int((st[:cap] != 0).sum())
    d = np.diff(st[:n])
    tags = st[cap + 1:cap + n]
    print("==", env, "N", N, "backward" if backward else "forward", "phases", n - 1, "total cycles", int(st[n - 1] - st[0]))
    print(" ".join(str(int(x)) for x in d))
    import collections
    agg = collections.OrderedDict()
    for t, x in zip(tags.tolist(), d.tolist()):
        agg.setdefault(t, []).append(x)
    names = {0: "io", 1: "fwd_kin", 2: "fwd_ext", 3: "fwd_tau", 4: "fwd_mass", 5: "fwd_solve", 6: "fwd_integ", 7: "bwd_joint",
             8: "bwd_ext", 9: "bwd_mass", 10: "bwd_bodies"}
    tot = collections.Counter()
    for t, xs in agg.items():
        tot[names.get(t // 100, "?")] += sum(xs)
        print("  %-10s phase %2d: n=%3d mean %6.0f cycles" % (names.get(t // 100, "?"), t % 100, len(xs), sum(xs) / len(xs)))
    print("  totals:", dict(tot))
