"""ctypes driver of the CPU oracle (oracle/libdsim_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from diffrl_amd.capi import make_desc
from diffrl_amd.template import ArticulationTemplate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
_lib = None


def oracle():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libdsim_oracle.so")
        src = os.path.join(ORACLE_DIR, "dsim_oracle.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        _lib = C.CDLL(so)
    return _lib


def golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as d:
        return {k: d[k] for k in d.files}


def template_from_golden(env):
    return ArticulationTemplate.from_reference_dump(golden(env + "_model"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def oracle_forward(t, q, qd, act, mact, dt, substeps, mm_freq, debug=False):
    desc, keep = make_desc(t)
    N = q.shape[0]
    q, qd, act = _c(q), _c(qd), _c(act)
    mact = _c(mact) if mact is not None else np.zeros((N, 0), np.float32)
    qo, qdo = np.zeros_like(q), np.zeros_like(qd)
    L, nd = t.n_links, t.n_qd
    dbg = {}
    if debug:
        dbg = dict(X_sc=np.zeros((N, L, 7), np.float32), X_sm=np.zeros((N, L, 7), np.float32),
                   S_s=np.zeros((N, nd, 6), np.float32), I_s=np.zeros((N, L, 6, 6), np.float32),
                   v_s=np.zeros((N, L, 6), np.float32), a_s=np.zeros((N, L, 6), np.float32),
                   f_s=np.zeros((N, L, 6), np.float32), ft_s=np.zeros((N, L, 6), np.float32),
                   tau=np.zeros((N, nd), np.float32), qdd=np.zeros((N, nd), np.float32),
                   H=np.zeros((N, nd, nd), np.float32), L=np.zeros((N, nd, nd), np.float32))
    names = ["X_sc", "X_sm", "S_s", "I_s", "v_s", "a_s", "f_s", "ft_s", "tau", "qdd", "H", "L"]
    fn = oracle().dsim_oracle_step_forward
    fn.restype = C.c_int
    rc = fn(C.byref(desc), C.c_int(N), _p(q), _p(qd), _p(act), _p(mact), C.c_float(dt), C.c_int(substeps),
            C.c_int(mm_freq), _p(qo), _p(qdo), *[_p(dbg.get(n)) for n in names])
    assert rc == 0
    return qo, qdo, dbg


def oracle_backward(t, q, qd, act, mact, dt, substeps, mm_freq, gq_out, gqd_out):
    desc, keep = make_desc(t)
    N = q.shape[0]
    q, qd, act, gq_out, gqd_out = _c(q), _c(qd), _c(act), _c(gq_out), _c(gqd_out)
    mact = _c(mact) if mact is not None else np.zeros((N, 0), np.float32)
    gq, gqd, ga, gm = np.zeros_like(q), np.zeros_like(qd), np.zeros_like(act), np.zeros_like(mact)
    qo, qdo = np.zeros_like(q), np.zeros_like(qd)
    fn = oracle().dsim_oracle_step_backward
    fn.restype = C.c_int
    rc = fn(C.byref(desc), C.c_int(N), _p(q), _p(qd), _p(act), _p(mact), C.c_float(dt), C.c_int(substeps),
            C.c_int(mm_freq), _p(gq_out), _p(gqd_out), _p(gq), _p(gqd), _p(ga), _p(gm), _p(qo), _p(qdo))
    assert rc == 0
    return dict(gq=gq, gqd=gqd, gact=ga, gmact=gm, q_out=qo, qd_out=qdo)


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def project_tangent(t, q, g):
    """Removes from a joint_q cotangent `g` the component along each unit-quaternion coordinate block
    (free joint q[3:7], ball joint q[0:4]).  The reference differentiates formulas that are only
    meaningful on |quat| = 1 literally, so its gradient has a radial part that depends on how the
    (mathematically identical) rotation formulas are written; that part is annihilated by the
    integrator's quaternion normalisation (sim.py:1552, 1616) in every upstream propagation, see
    DESIGN.md "Quaternion radial component"."""
    g = np.array(g, np.float64).reshape(-1, t.n_q)
    q = np.asarray(q, np.float64).reshape(-1, t.n_q)
    for i in range(t.n_links):
        ty, cs = int(t.joint_type[i]), int(t.joint_q_start[i])
        if ty == 4:
            sl = slice(cs + 3, cs + 7)
        elif ty == 2:
            sl = slice(cs, cs + 4)
        else:
            continue
        u = q[:, sl] / np.linalg.norm(q[:, sl], axis=1, keepdims=True)
        g[:, sl] -= u * (u * g[:, sl]).sum(1, keepdims=True)
    return g


def step_grad_tolerance(t, q, qd, act, mact, dt, substeps, mm_freq, gq_out, gqd_out, measured, ref=None, budget=0, what=""):
    """Tolerance of an env-step gradient comparison: 1e-3 unless the REFERENCE-order gradient is itself
    more sensitive than that at these inputs -- the oracle's gradients recomputed from coordinates moved by 1e-7 (relative)
    bound what any re-association of the fp32 arithmetic can promise (contacts and joint limits switch at thresholds).
    measured: dict name -> error of this implementation; returns dict name -> tolerance.  Every quantity that needs the probe
    is recorded loudly and counted against the calling test's `budget` of probed cases; the accepted error never exceeds
    the step-level ceiling (tests/probe_ledger.py)."""
    import probe_ledger
    tol = {k: 1e-3 for k in measured}
    if any(measured[k] >= 1e-3 for k in measured):
        ref = ref or oracle_backward(t, q, qd, act, mact, dt, substeps, mm_freq, gq_out, gqd_out)
        rng = np.random.default_rng(0)
        qp = (np.asarray(q, np.float64) * (1.0 + 1e-7 * rng.normal(size=np.shape(q)))).astype(np.float32)
        pr = oracle_backward(t, qp, qd, act, mact, dt, substeps, mm_freq, gq_out, gqd_out)
        for k in tol:
            if measured[k] < 1e-3:
                continue
            a, b = (project_tangent(t, q, pr[k]), project_tangent(t, q, ref[k])) if k == "gq" else (pr[k], ref[k])
            tol[k] = probe_ledger.accept("step", measured[k], relerr(a, b), budget, "%s %s" % (what, k))
    return tol


def radial_split(t, q, ours, ref):
    """UN-projected comparison of a joint_q cotangent with the reference's at the operator boundary (include/dsim.h,
    dsim_step_backward).  Returns (max |radial part of ours|, max |ours - ref + radial part of ref| over the quaternion blocks,
    max |ours - ref| over every other coordinate, max |radial part of ref|), all relative to max |ref|: the first three must
    vanish -- the whole un-projected difference IS the reference's radial part -- the fourth is its reported size."""
    ours, ref, q = np.array(ours, np.float64), np.array(ref, np.float64), np.asarray(q, np.float64)
    scale = np.abs(ref).max()
    diff = ours - ref
    own_rad = resid = rad_ref = 0.0
    blocks = 0
    for i in range(t.n_links):
        ty, cs = int(t.joint_type[i]), int(t.joint_q_start[i])
        sl = slice(cs + 3, cs + 7) if ty == 4 else (slice(cs, cs + 4) if ty == 2 else None)
        if sl is None:
            continue
        blocks += 1
        u = q[:, sl] / np.linalg.norm(q[:, sl], axis=1, keepdims=True)
        ref_rad = (u * ref[:, sl]).sum(1)
        own_rad = max(own_rad, np.abs((u * ours[:, sl]).sum(1)).max())
        resid = max(resid, np.abs(diff[:, sl] + u * ref_rad[:, None]).max())
        rad_ref = max(rad_ref, np.abs(ref_rad).max())
        diff[:, sl] = 0.0
    assert blocks
    return own_rad / scale, resid / scale, np.abs(diff).max() / scale, rad_ref / scale
