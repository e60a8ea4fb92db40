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
        subprocess.check_call(["make", "-C", EMU_DIR, "-s"])
        _lib = C.CDLL(os.path.join(EMU_DIR, "libdsim_emu.so"))
    return _lib


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
    desc, keep = make_desc(t)
    names = _off_names()
    off = np.zeros(len(names) + 8, np.int32)
    dims = np.zeros(8, np.int32)
    n = emu().dsim_emu_layout(C.byref(desc), _p(off), C.c_int(off.size), _p(dims))
    assert n == len(names), (n, len(names))
    return dict(zip(names, off[:n].tolist())), dict(zip("L nq nd C M W NS D".split(), dims.tolist()))


def substep_image(t, q, qd, act, mact, h):
    """One substep (mass refresh, no integrate) for one env; returns dict name -> LDS slice getter."""
    desc, keep = make_desc(t)
    off, dims = layout(t)
    img = np.zeros(off["total_words"], np.float32)
    mact = _c(mact) if mact is not None else np.zeros(0, np.float32)
    rc = emu().dsim_emu_substep_image(C.byref(desc), _p(_c(q)), _p(_c(qd)), _p(_c(act)), _p(mact), C.c_float(h), _p(img))
    assert rc == 0
    return img, off, dims


def emu_forward(t, q, qd, act, mact, dt, substeps, mm_freq, want_ckpt=False):
    desc, keep = make_desc(t)
    N = q.shape[0]
    q, qd, act = _c(q), _c(qd), _c(act)
    mact = _c(mact) if mact is not None else np.zeros((N, 0), np.float32)
    qo, qdo = np.zeros_like(q), np.zeros_like(qd)
    ck = np.zeros((N, substeps, t.n_q + t.n_qd), np.float32) if want_ckpt else None
    rc = emu().dsim_emu_step_forward(C.byref(desc), C.c_int(N), _p(q), _p(qd), _p(act), _p(mact), C.c_float(dt),
                                     C.c_int(substeps), C.c_int(mm_freq), _p(qo), _p(qdo), _p(ck))
    assert rc == 0
    return qo, qdo, ck


def emu_backward(t, ckpt, act, mact, dt, substeps, mm_freq, gq_out, gqd_out):
    desc, keep = make_desc(t)
    N = act.shape[0]
    ckpt, act, gq_out, gqd_out = _c(ckpt), _c(act), _c(gq_out), _c(gqd_out)
    mact = _c(mact) if mact is not None else np.zeros((N, 0), np.float32)
    gq, gqd, ga, gm = np.zeros_like(gq_out), np.zeros_like(gqd_out), np.zeros_like(act), np.zeros_like(mact)
    rc = emu().dsim_emu_step_backward(C.byref(desc), C.c_int(N), _p(ckpt), _p(act), _p(mact), C.c_float(dt),
                                      C.c_int(substeps), C.c_int(mm_freq), _p(gq_out), _p(gqd_out), _p(gq), _p(gqd),
                                      _p(ga), _p(gm))
    assert rc == 0
    return dict(gq=gq, gqd=gqd, gact=ga, gmact=gm)
