"""ctypes driver of the lane-serial host build of the kernel phase code (tests/emu).  Test-only."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from diffrl_amd.capi import make_desc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
_lib = None


def emu():
    global _lib
    if _lib is None:
        # DSIM_EMU_LIB: another build of the harness (tests/test_probe_ledger.py runs the parity tests against the one with
        # an injected adjoint error, libdsim_emu_inject.so)
        name = os.environ.get("DSIM_EMU_LIB") or "libdsim_emu.so"
        subprocess.check_call(["make", "-C", EMU_DIR, "-s", name] if name != "libdsim_emu.so" else ["make", "-C", EMU_DIR, "-s"])
        _lib = C.CDLL(os.path.join(EMU_DIR, name))
    return _lib


_lay = None


def layout_lib():
    """the layout builder alone (tests/emu/dsim_layout_tool.cpp): does not depend on the generated static layouts"""
    global _lay
    if _lay is None:
        subprocess.check_call(["make", "-C", EMU_DIR, "-s", "libdsim_layout.so"])
        _lay = C.CDLL(os.path.join(EMU_DIR, "libdsim_layout.so"))
    return _lay


def _off_names():
    src = open(os.path.join(ROOT, "diffrl_amd", "csrc", "dsim_layout.hpp")).read()
    body = src[src.index("struct DsimOff {"):]
    body = body[:body.index("};")]
    body = re.sub(r"//.*", "", body)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if stmt.startswith("struct"):
            stmt = stmt[stmt.index("{") + 1:].strip()
        if stmt.startswith("int "):
            names += [n.strip() for n in stmt[4:].split(",")]
    return names


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def layout(t):
    """(offsets, dims) of template t from the library's layout builder (the product's host-only build of it)"""
    from diffrl_amd import specialise
    return specialise.layout(t)


SCAN_MIN_DEPTH = 5   # dsim_core.hpp: DSIM_SCAN_MIN_DEPTH


def reorders(t):
    """True if the one-wave SPECIALISED kernels of this model associate sums / products differently from the generic
    kernels (same terms, different order): trunk decomposition of the subtree sums (dsim_trunk_sum) and / or the log-depth
    kinematics of deep trees (dsim_fwd_kinematics_scan).  Such pairs agree to a tolerance, all others bit for bit."""
    d = layout(t)[1]
    scan = d["D"] >= SCAN_MIN_DEPTH and d["L"] <= 64 and d["nd"] <= 64 and d["C"] <= 64 and d["NS"] <= 64
    # dsim_core.hpp: DsimRowTree (body-level adjoint with subtree sums by lane shifts; models with muscles on their first wave)
    rowtree = d["RT_N"] > 0 and (d["flags"] & 1) and (d["NS"] > 0 or d["CBMAX"] <= 12)
    return d["NT"] > 0 or scan or rowtree


def substep_image(t, q, qd, act, mact, h):
    """One substep (mass refresh, no integrate) for one env; returns dict name -> LDS slice getter."""
    desc, keep = make_desc(t)
    off, dims = layout(t)
    img = np.zeros(off["total_words"], np.float32)
    mact = _c(mact) if mact is not None else np.zeros(0, np.float32)
    rc = emu().dsim_emu_substep_image(C.byref(desc), _p(_c(q)), _p(_c(qd)), _p(_c(act)), _p(mact), C.c_float(h), _p(img))
    assert rc == 0
    return img, off, dims


def ckpt_floats(t, substeps, mm_freq):
    desc, keep = make_desc(t)
    fn = emu().dsim_emu_ckpt_floats
    fn.restype = C.c_longlong
    return int(fn(C.byref(desc), C.c_int(substeps), C.c_int(mm_freq)))


def emu_forward(t, q, qd, act, mact, dt, substeps, mm_freq, want_ckpt=False):
    desc, keep = make_desc(t)
    N = q.shape[0]
    q, qd, act = _c(q), _c(qd), _c(act)
    mact = _c(mact) if mact is not None else np.zeros((N, 0), np.float32)
    qo, qdo = np.zeros_like(q), np.zeros_like(qd)
    ck = np.zeros((N, ckpt_floats(t, substeps, mm_freq)), np.float32) if want_ckpt else None
    rc = emu().dsim_emu_step_forward(C.byref(desc), C.c_int(N), _p(q), _p(qd), _p(act), _p(mact), C.c_float(dt),
                                     C.c_int(substeps), C.c_int(mm_freq), _p(qo), _p(qdo), _p(ck))
    assert rc == 0
    return qo, qdo, ck


def emu_backward(t, ckpt, act, mact, dt, substeps, mm_freq, gq_out, gqd_out, literal=False, lit_out=None):
    """literal: dsim_step_backward_literal -- the quaternion blocks of gq carry the reference's radial component too"""
    desc, keep = make_desc(t)
    N = act.shape[0]
    ckpt, act, gq_out, gqd_out = _c(ckpt), _c(act), _c(gq_out), _c(gqd_out)
    mact = _c(mact) if mact is not None else np.zeros((N, 0), np.float32)
    gq, gqd, ga, gm = np.zeros_like(gq_out), np.zeros_like(gqd_out), np.zeros_like(act), np.zeros_like(mact)
    fn = emu().dsim_emu_step_backward_literal if literal else emu().dsim_emu_step_backward
    extra = (_p(lit_out),) if literal else ()
    rc = fn(C.byref(desc), C.c_int(N), _p(ckpt), _p(act), _p(mact), C.c_float(dt),
            C.c_int(substeps), C.c_int(mm_freq), _p(gq_out), _p(gqd_out), _p(gq), _p(gqd),
            _p(ga), _p(gm), *extra)
    assert rc == 0
    return dict(gq=gq, gqd=gqd, gact=ga, gmact=gm)


# ---- fused env surface ------------------------------------------------------------------------------
def env_spec_for(env_name, t):
    """(capi.EnvSpec, keepalive) with the constants of diffrl_amd.envs.<env> (host act_scale array)."""
    import math

    from diffrl_amd import capi
    from diffrl_amd.dflex import util as U
    if env_name == "cartpole":
        sc = np.full(1, 1000.0, np.float32)
        return capi.make_env_spec(capi.ENV_CARTPOLE, capi.REW_CARTPOLE, 1, 5, sc.ctypes.data, action_penalty=0.0,
                                  cartpole_penalties=(1.0, 0.1, 0.05, 0.1)), sc
    if env_name in ("hopper", "cheetah"):
        na = 3 if env_name == "hopper" else 6
        sc = np.full(na, 200.0, np.float32)
        if env_name == "hopper":
            return capi.make_env_spec(capi.ENV_PLANAR, capi.REW_HOPPER, 3, 11, sc.ctypes.data, act_offset=3,
                                      action_penalty=-0.1, termination_height=-0.45, termination_tolerance=0.15,
                                      height_rew_scale=1.0, cartpole_penalties=(math.pi / 6.0, 0.0, 0.0, 0.0)), sc
        return capi.make_env_spec(capi.ENV_PLANAR, capi.REW_CHEETAH, 6, 17, sc.ctypes.data, act_offset=3,
                                  action_penalty=-0.1), sc
    if env_name == "ant":
        sr, h, tgt = U.quat_from_axis_angle((1.0, 0.0, 0.0), -math.pi * 0.5), 0.75, 10000.0
        sc = np.full(8, 200.0, np.float32)
        kw = dict(rew=capi.REW_ANT, n_act=8, n_obs=37, off=6, mus=False, oa=True, th=0.27, tol=0.0, hs=0.0, pen=0.0)
    elif env_name == "humanoid":
        sr, h, tgt = U.quat_from_axis_angle((1.0, 0.0, 0.0), -math.pi * 0.5), 1.35, 200.0
        ms = [200, 200, 200, 200, 200, 600, 400, 100, 100, 200, 200, 600, 400, 100, 100, 100, 100, 200, 100, 100, 200]
        sc = (0.35 * np.array(ms, np.float64)).astype(np.float32)
        kw = dict(rew=capi.REW_HUMANOID, n_act=21, n_obs=76, off=6, mus=False, oa=True, th=0.74, tol=0.1, hs=10.0,
                  pen=-0.002)
    else:
        sr, h, tgt = U.quat_from_axis_angle((0.0, 1.0, 0.0), math.pi * 0.5), 1.0, 10000.0
        sc = np.ascontiguousarray(t.extras["muscle_strengths"], np.float32)
        kw = dict(rew=capi.REW_SNU, n_act=152, n_obs=53, off=0, mus=True, oa=False, th=0.46, tol=0.05, hs=4.0, pen=-0.001)
    isr = (-sr[0], -sr[1], -sr[2], sr[3])
    isr = tuple(float(np.float32(v)) for v in isr)
    spec = capi.make_env_spec(capi.ENV_LOCOMOTION, kw["rew"], kw["n_act"], kw["n_obs"], sc.ctypes.data,
                              act_offset=kw["off"], act_muscle=kw["mus"], obs_actions=kw["oa"], inv_start_rot=isr,
                              target_xz=(tgt, 0.0), termination_height=kw["th"], termination_tolerance=kw["tol"],
                              height_rew_scale=kw["hs"], action_penalty=kw["pen"], joint_vel_obs_scaling=0.1)
    return spec, sc


def make_episode(progress, done, obs_before, reset_q, reset_qd, reset_count, episode_length, height_terminate,
                 check_invalid, noise_q=None, noise_qd=None, noise_angle=0.0, seed=0):
    """capi.Episode over numpy arrays (int64 progress/done, float32 pool [K][N][.], int32 reset_count)"""
    from diffrl_amd import capi
    ep = capi.Episode()
    ep.progress, ep.done = progress.ctypes.data, done.ctypes.data
    ep.obs_before_reset = obs_before.ctypes.data if obs_before is not None else None
    ep.reset_q, ep.reset_qd, ep.reset_count = reset_q.ctypes.data, reset_qd.ctypes.data, reset_count.ctypes.data
    ep.reset_pool, ep.episode_length = int(reset_q.shape[0]), int(episode_length)
    ep.height_terminate, ep.check_invalid = int(bool(height_terminate)), int(bool(check_invalid))
    ep.noise_q = noise_q.ctypes.data if noise_q is not None else None
    ep.noise_qd = noise_qd.ctypes.data if noise_qd is not None else None
    ep.noise_angle, ep.seed = float(noise_angle), int(seed)
    return ep


def emu_env_forward(t, spec, q, qd, actions, dt, substeps, mm_freq, episode=None):
    desc, keep = make_desc(t)
    N = q.shape[0]
    q, qd, actions = _c(q), _c(qd), _c(actions)
    qo, qdo = np.zeros_like(q), np.zeros_like(qd)
    obs, rew = np.zeros((N, spec.n_obs), np.float32), np.zeros(N, np.float32)
    ck = np.zeros((N, ckpt_floats(t, substeps, mm_freq)), np.float32)
    rc = emu().dsim_emu_env_forward(C.byref(desc), C.byref(spec), C.c_int(N), _p(q), _p(qd), _p(actions), C.c_float(dt),
                                    C.c_int(substeps), C.c_int(mm_freq), _p(qo), _p(qdo), _p(obs), _p(rew), _p(ck),
                                    C.byref(episode) if episode is not None else None)
    assert rc == 0
    return qo, qdo, obs, rew, ck


def emu_env_backward(t, spec, ck, actions, dt, substeps, mm_freq, gq_out, gqd_out, gobs, grew, gobs_before=None):
    desc, keep = make_desc(t)
    N = actions.shape[0]
    ck, actions = _c(ck), _c(actions)
    g = [_c(a) if a is not None else None for a in (gq_out, gqd_out, gobs, grew, gobs_before)]
    nq, nd = t.n_q, t.n_qd
    gq, gqd, ga = np.zeros((N, nq), np.float32), np.zeros((N, nd), np.float32), np.zeros_like(actions)
    po = lambda a: _p(a) if a is not None else None  # noqa: E731
    rc = emu().dsim_emu_env_backward(C.byref(desc), C.byref(spec), C.c_int(N), _p(ck), _p(actions), C.c_float(dt),
                                     C.c_int(substeps), C.c_int(mm_freq), po(g[0]), po(g[1]), po(g[2]), po(g[3]),
                                     po(g[4]), _p(gq), _p(gqd), _p(ga))
    assert rc == 0
    return gq, gqd, ga
