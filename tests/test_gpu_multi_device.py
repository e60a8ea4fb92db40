"""-m gpu: one process driving several GPUs ("one model-template handle + one stream (or process) per
GPU").  A model belongs to the device it was created on; the multi-device cases need a lease with >= 2 GPUs and skip LOUDLY
on a 1-GPU box (the driver's 8-GPU scaling run uses one process per GPU, bench.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle_lib import golden, template_from_golden

pytestmark = pytest.mark.gpu


def _inputs(dev, n=64):
    g = golden("ant_step")
    reps = n // g["q_in"].shape[0] + 1
    T = lambda a: torch.tensor(np.tile(a, (reps, 1))[:n], device=dev).reshape(-1)  # noqa: E731
    return g, T(g["q_in"]), T(g["qd_in"]), T(g["act_in"])


def test_model_records_its_device():
    from diffrl_amd.engine import Engine
    eng = Engine(template_from_golden("ant"), "cuda")          # no index: the current device
    assert eng.device == torch.device("cuda", torch.cuda.current_device())
    assert int(eng._lib.dsim_model_device(eng._h)) == torch.cuda.current_device()


def test_checkpoint_of_another_geometry_is_refused():
    """a checkpoint written with one (substeps, mm_freq, mode) must not be consumed with another: the row stride differs"""
    from diffrl_amd import capi
    from diffrl_amd.engine import Engine
    dev = torch.device("cuda:0")
    g, q, qd, act = _inputs(dev, 8)
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    full, lean = Engine(template_from_golden("ant"), dev, ckpt_mode="full"), Engine(template_from_golden("ant"), dev, ckpt_mode="lean")
    _, _, ck = full.forward(q, qd, act, None, dt, S, mm, True)
    gq, gqd = torch.randn_like(q), torch.randn_like(qd)
    full.backward(ck, act, None, dt, S, mm, gq, gqd)
    with pytest.raises(capi.DsimError):
        lean.backward(ck, act, None, dt, S, mm, gq, gqd)
    with pytest.raises(capi.DsimError):
        full.backward(ck, act, None, dt, S, 1, gq, gqd)


def _need_two():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("MULTI-GPU CASE NOT RUN: this lease exposes %d GPU; two engines on two devices in one process need >= 2" % n)


def test_two_engines_on_two_devices_match_one_device():
    """two Engines in ONE process, on cuda:0 and cuda:1, each with its half of the environments: bit-equal to the whole batch
    on one device (environments are independent; no collective), forward, adjoint and the fused env surface"""
    _need_two()
    from diffrl_amd.engine import Engine
    t = template_from_golden("ant")
    g, q, qd, act = _inputs(torch.device("cuda:0"), 64)
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    gq, gqd = torch.randn_like(q), torch.randn_like(qd)
    e0 = Engine(t, "cuda:0")
    qo, qdo, ck = e0.forward(q, qd, act, None, dt, S, mm, True)
    ref = (qo, qdo) + tuple(e0.backward(ck, act, None, dt, S, mm, gq, gqd)[:3])
    outs = []
    for k, dev in enumerate(("cuda:0", "cuda:1")):
        eng = Engine(t, dev)
        assert int(eng._lib.dsim_model_device(eng._h)) == k
        sl = lambda x, w: x.view(64, w)[32 * k:32 * (k + 1)].reshape(-1).to(dev)  # noqa: E731
        a, b, c, d, e = sl(q, 15), sl(qd, 14), sl(act, 14), sl(gq, 15), sl(gqd, 14)
        qo_k, qdo_k, ck_k = eng.forward(a, b, c, None, dt, S, mm, True)
        outs.append((qo_k, qdo_k) + tuple(eng.backward(ck_k, c, None, dt, S, mm, d, e)[:3]))
    torch.cuda.synchronize(0)
    torch.cuda.synchronize(1)
    for i, w in enumerate((15, 14, 15, 14, 14)):
        got = torch.cat([outs[0][i].cpu().view(32, w), outs[1][i].cpu().view(32, w)])
        assert torch.equal(got, ref[i].cpu().view(64, w)), i


def test_call_with_another_current_device_is_refused():
    """the C ABI checks the current device against the model's: a launch elsewhere would read constants that are not there"""
    _need_two()
    from diffrl_amd import capi
    from diffrl_amd.engine import Engine
    eng = Engine(template_from_golden("ant"), "cuda:1")
    g, q, qd, act = _inputs(torch.device("cuda:1"), 8)
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    qo, qdo = torch.empty_like(q), torch.empty_like(qd)
    p = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    with torch.cuda.device(0):
        rc = eng._lib.dsim_step_forward(eng._h, 8, p(q), p(qd), p(act), None, C.c_float(dt), S, mm, p(qo), p(qdo), None, None)
    assert rc == capi.ERR_INVALID and b"device" in eng._lib.dsim_last_error()
    with torch.cuda.device(1):
        rc = eng._lib.dsim_step_forward(eng._h, 8, p(q), p(qd), p(act), None, C.c_float(dt), S, mm, p(qo), p(qdo), None, None)
    assert rc == 0
