"""-m gpu: the real HIP kernels, through the C ABI, against (a) the golden vectors generated from the
reference and (b) the CPU oracle on larger seeded batches.  Tolerances (stated, fp32):
one env-step state 1e-4, one env-step gradients 1e-3 max-norm relative."""
import numpy as np
import pytest
import torch

from oracle_lib import golden, oracle_backward, project_tangent, relerr, template_from_golden

pytestmark = pytest.mark.gpu
ENVS = ["cartpole", "ant", "humanoid", "snu", "hopper", "cheetah"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _engine(env, dev):
    from diffrl_amd.engine import Engine
    t = template_from_golden(env)
    return t, Engine(t, dev)


def _run(eng, dev, q, qd, act, mact, dt, S, mm, gq, gqd):
    tq = torch.tensor(q, device=dev).reshape(-1)
    tqd = torch.tensor(qd, device=dev).reshape(-1)
    ta = torch.tensor(act, device=dev).reshape(-1)
    tm = torch.tensor(mact, device=dev).reshape(-1) if mact is not None else None
    qo, qdo, ck = eng.forward(tq, tqd, ta, tm, dt, S, mm, True)
    r = eng.backward(ck, ta, tm, dt, S, mm, torch.tensor(gq, device=dev).reshape(-1),
                     torch.tensor(gqd, device=dev).reshape(-1))
    torch.cuda.synchronize()
    n = q.shape[0]
    out = dict(q=qo.cpu().numpy().reshape(n, -1), qd=qdo.cpu().numpy().reshape(n, -1),
               gq=r[0].cpu().numpy().reshape(n, -1), gqd=r[1].cpu().numpy().reshape(n, -1),
               gact=r[2].cpu().numpy().reshape(n, -1), ckpt=ck.cpu().numpy())
    if r[3] is not None:
        out["gmact"] = r[3].cpu().numpy().reshape(n, -1)
    return out


@pytest.mark.parametrize("env", ENVS)
def test_step_vs_reference_golden(env, dev):
    t, eng = _engine(env, dev)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    r = _run(eng, dev, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    assert relerr(r["q"], g["q_out"]) < 1e-4
    assert relerr(r["qd"], g["qd_out"]) < 1e-4
    assert np.array_equal(r["ckpt"][:, :t.n_q], g["q_in"])  # every checkpoint row starts with the substep's input q
    assert relerr(project_tangent(t, g["q_in"], r["gq"]), project_tangent(t, g["q_in"], g["gq_in"])) < 1e-3
    assert relerr(r["gqd"], g["gqd_in"]) < 1e-3
    if "gact_in" in g:
        assert relerr(r["gact"], g["gact_in"]) < 1e-3
    else:
        assert relerr(r["gmact"], g["gmuscle_act_in"]) < 1e-3


@pytest.mark.parametrize("env", ["ant", "humanoid", "snu"])
def test_unprojected_gq_differs_from_the_reference_by_its_radial_part_only(env, dev):
    """include/dsim.h (dsim_step_backward) states ONE deviation from the reference's adjoint at the operator boundary: the
    cotangent of a quaternion coordinate block has no component along the quaternion itself.  Asserted here on what the HIP
    adjoint kernel returns, WITHOUT any projection: (i) that component is zero, (ii) the difference to the reference's
    recording is exactly the reference's own radial component, (iii) every other coordinate agrees as it stands; the size of
    the reference's radial part is printed (32 % / 16 % / 8 % of max |gq| on these recordings)."""
    from oracle_lib import radial_split
    t, eng = _engine(env, dev)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    r = _run(eng, dev, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    own_rad, resid, other, rad_ref = radial_split(t, g["q_in"], r["gq"], g["gq_in"])
    print("%s: radial part of this adjoint %.1e, residual of (ours - ref + ref's radial part) %.1e, other coordinates %.1e; "
          "reference's radial part %.3f of max |gq|" % (env, own_rad, resid, other, rad_ref))
    assert own_rad < 1e-4 and resid < 1e-3 and other < 1e-3
    assert rad_ref > 0.01      # (the recordings do exercise it)


@pytest.mark.parametrize("env", ["ant", "humanoid", "snu"])
def test_literal_backward_equals_the_reference_unprojected(env, dev):
    """SURVEY 8(a) row SimulateFunc + Tape reverse sweep, literally: dsim_step_backward_literal (Engine.backward(literal=True),
    dflex.config.literal_quat_grad) returns the reference's joint_q cotangent INCLUDING the component along each quaternion
    (quat.h:232-288, spatial.h:740-798) -- compared with the reference's recording as it stands, no projection anywhere,
    stated tolerance 1e-3 (measured ~1e-6); the other outputs are those of the plain call, bit for bit"""
    t, eng = _engine(env, dev)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    T = lambda a: torch.tensor(a, device=dev).reshape(-1)  # noqa: E731
    mact = T(g["muscle_act_in"]) if "muscle_act_in" in g else None
    qo, qdo, ck = eng.forward(T(g["q_in"]), T(g["qd_in"]), T(g["act_in"]), mact, dt, S, mm, True)
    plain = eng.backward(ck, T(g["act_in"]), mact, dt, S, mm, T(g["gq_out"]), T(g["gqd_out"]))
    lit = eng.backward(ck, T(g["act_in"]), mact, dt, S, mm, T(g["gq_out"]), T(g["gqd_out"]), literal=True)
    torch.cuda.synchronize()
    n = g["q_in"].shape[0]
    e = relerr(lit[0].cpu().numpy().reshape(n, -1), g["gq_in"])
    print("%s: un-projected |gq_literal - gq_reference| / max|gq_reference| = %.2e (plain call: %.2e)"
          % (env, e, relerr(plain[0].cpu().numpy().reshape(n, -1), g["gq_in"])))
    assert e < 1e-3
    for a, b in zip(plain[1:], lit[1:]):
        assert (a is None and b is None) or torch.equal(a, b)
    # the two calls differ in the quaternion blocks only, and there by a multiple of the quaternion
    d = (lit[0] - plain[0]).cpu().numpy().reshape(n, -1)
    for i in range(t.n_links):
        ty, cs = int(t.joint_type[i]), int(t.joint_q_start[i])
        sl = slice(cs + 3, cs + 7) if ty == 4 else (slice(cs, cs + 4) if ty == 2 else None)
        if sl is None:
            continue
        u = g["q_in"][:, sl]
        rho = (d[:, sl] * u).sum(1, keepdims=True)
        assert np.abs(d[:, sl] - rho * u).max() <= 1e-5 * max(1.0, np.abs(d[:, sl]).max())
        d[:, sl] = 0.0
    assert np.abs(d).max() == 0.0


@pytest.mark.parametrize("env,n", [("ant", 96), ("humanoid", 12), ("snu", 8)])
def test_literal_backward_vs_oracle_batch(env, n, dev):
    """seeded perturbations of the golden states, random cotangents, a step geometry whose first mass-matrix group is shorter than
    the step: the literal call against the scalar oracle's reference-order adjoint, un-projected"""
    t, eng = _engine(env, dev)
    g = golden(env + "_step")
    S, mm, dt = 6, 4, 6 * float(g["dt"]) / int(g["substeps"])
    rng = np.random.default_rng(9)
    idx = rng.integers(0, g["q_in"].shape[0], n)
    q = g["q_in"][idx] + rng.normal(0, 0.01, (n, t.n_q)).astype(np.float32)
    for i in range(t.n_links):
        ty, cs = int(t.joint_type[i]), int(t.joint_q_start[i])
        sl = slice(cs + 3, cs + 7) if ty == 4 else (slice(cs, cs + 4) if ty == 2 else None)
        if sl is not None:
            q[:, sl] /= np.linalg.norm(q[:, sl], axis=1, keepdims=True)
    qd = g["qd_in"][idx] + rng.normal(0, 0.05, (n, t.n_qd)).astype(np.float32)
    act = g["act_in"][idx] * rng.uniform(0.5, 1.0, (n, 1)).astype(np.float32)
    mact = g["muscle_act_in"][idx] * rng.uniform(0.5, 1.0, (n, 1)).astype(np.float32) if "muscle_act_in" in g else None
    gq, gqd = rng.normal(0, 1, (n, t.n_q)).astype(np.float32), rng.normal(0, 1, (n, t.n_qd)).astype(np.float32)
    o = oracle_backward(t, q, qd, act, mact, dt, S, mm, gq, gqd)
    T = lambda a: torch.tensor(a, device=dev).reshape(-1)  # noqa: E731
    qo, qdo, ck = eng.forward(T(q), T(qd), T(act), T(mact) if mact is not None else None, dt, S, mm, True)
    r = eng.backward(ck, T(act), T(mact) if mact is not None else None, dt, S, mm, T(gq), T(gqd), literal=True)
    torch.cuda.synchronize()
    assert relerr(r[0].cpu().numpy().reshape(n, -1), o["gq"]) < 1e-3
    assert relerr(r[1].cpu().numpy().reshape(n, -1), o["gqd"]) < 1e-3


def test_simstep_honours_literal_quat_grad(dev):
    """dflex.config.literal_quat_grad: SemiImplicitIntegrator.forward's autograd node returns the literal cotangent"""
    from diffrl_amd.dflex import config
    from diffrl_amd.engine import SimStep
    t, eng = _engine("ant", dev)
    g = golden("ant_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    n = g["q_in"].shape[0]
    out = {}
    for flag in (False, True):
        config.literal_quat_grad = flag
        try:
            q = torch.tensor(g["q_in"], device=dev).reshape(-1).requires_grad_(True)
            qd = torch.tensor(g["qd_in"], device=dev).reshape(-1).requires_grad_(True)
            a = torch.tensor(g["act_in"], device=dev).reshape(-1).requires_grad_(True)
            qo, qdo = SimStep.apply(eng, dt, S, mm, q, qd, a, None)
            ((qo * torch.tensor(g["gq_out"], device=dev).reshape(-1)).sum() + (qdo * torch.tensor(g["gqd_out"], device=dev).reshape(-1)).sum()).backward()
            out[flag] = q.grad.cpu().numpy().reshape(n, -1)
        finally:
            config.literal_quat_grad = False
    assert relerr(out[True], g["gq_in"]) < 1e-3 and relerr(out[False], g["gq_in"]) > 0.05


@pytest.mark.parametrize("env,n", [("cartpole", 256), ("ant", 192), ("humanoid", 24), ("snu", 16)])
def test_step_vs_oracle_batch(env, n, dev):
    """seeded perturbations of the golden states; sizes the scalar oracle finishes in seconds"""
    t, eng = _engine(env, dev)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    rng = np.random.default_rng(5)
    idx = rng.integers(0, g["q_in"].shape[0], n)
    q = g["q_in"][idx] + rng.normal(0, 0.01, (n, t.n_q)).astype(np.float32)
    for i in range(t.n_links):  # keep unit quaternions
        ty, cs = int(t.joint_type[i]), int(t.joint_q_start[i])
        sl = slice(cs + 3, cs + 7) if ty == 4 else (slice(cs, cs + 4) if ty == 2 else None)
        if sl is not None:
            q[:, sl] /= np.linalg.norm(q[:, sl], axis=1, keepdims=True)
    qd = g["qd_in"][idx] + rng.normal(0, 0.05, (n, t.n_qd)).astype(np.float32)
    act = g["act_in"][idx] * rng.uniform(0.5, 1.0, (n, 1)).astype(np.float32)
    mact = g["muscle_act_in"][idx] * rng.uniform(0.5, 1.0, (n, 1)).astype(np.float32) if "muscle_act_in" in g else None
    gq = rng.normal(0, 1, (n, t.n_q)).astype(np.float32)
    gqd = rng.normal(0, 1, (n, t.n_qd)).astype(np.float32)
    r = _run(eng, dev, q, qd, act, mact, dt, S, mm, gq, gqd)
    o = oracle_backward(t, q, qd, act, mact, dt, S, mm, gq, gqd)
    assert relerr(r["q"], o["q_out"]) < 1e-4
    assert relerr(r["qd"], o["qd_out"]) < 1e-4
    assert relerr(project_tangent(t, q, r["gq"]), project_tangent(t, q, o["gq"])) < 1e-3
    assert relerr(r["gqd"], o["gqd"]) < 1e-3
    if mact is None:
        assert relerr(r["gact"], o["gact"]) < 1e-3
    else:
        assert relerr(r["gmact"], o["gmact"]) < 1e-3


def test_determinism_and_inplace(dev):
    """fixed reduction order per env: two launches are bit-identical; q_out may alias q_in"""
    t, eng = _engine("ant", dev)
    g = golden("ant_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    q = torch.tensor(np.tile(g["q_in"], (100, 1)), device=dev).reshape(-1)
    qd = torch.tensor(np.tile(g["qd_in"], (100, 1)), device=dev).reshape(-1)
    a = torch.tensor(np.tile(g["act_in"], (100, 1)), device=dev).reshape(-1)
    r1 = eng.forward(q, qd, a, None, dt, S, mm, True)
    r2 = eng.forward(q, qd, a, None, dt, S, mm, True)
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1])
    assert torch.equal(r1[2][:, :t.n_q], r2[2][:, :t.n_q])  # (rows also carry alignment padding that is never read)
    gq = torch.randn_like(q)
    gqd = torch.randn_like(qd)
    b1 = eng.backward(r1[2], a, None, dt, S, mm, gq, gqd)
    b2 = eng.backward(r1[2], a, None, dt, S, mm, gq, gqd)
    assert all(torch.equal(x, y) for x, y in zip(b1[:3], b2[:3]))
    # all 100 replicas of the same env agree bit-for-bit
    assert torch.equal(r1[0].view(100, -1, t.n_q)[0], r1[0].view(100, -1, t.n_q)[57])


def test_full_size_properties(dev):
    """BASELINE config size (Ant 1024 envs): size-independent properties instead of the scalar oracle:
    linearity of the adjoint in its seed, zero seed -> zero gradient, replicas identical, and the no-grad
    path (ckpt = NULL) returns the same states."""
    t, eng = _engine("ant", dev)
    g = golden("ant_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    reps = 1024 // g["q_in"].shape[0] + 1
    q = torch.tensor(np.tile(g["q_in"], (reps, 1))[:1024], device=dev).reshape(-1)
    qd = torch.tensor(np.tile(g["qd_in"], (reps, 1))[:1024], device=dev).reshape(-1)
    a = torch.tensor(np.tile(g["act_in"], (reps, 1))[:1024], device=dev).reshape(-1)
    qo, qdo, ck = eng.forward(q, qd, a, None, dt, S, mm, True)
    qo2, qdo2, none = eng.forward(q, qd, a, None, dt, S, mm, False)
    assert none is None and torch.equal(qo, qo2) and torch.equal(qdo, qdo2)
    g1q, g1d = torch.randn_like(q), torch.randn_like(qd)
    g2q, g2d = torch.randn_like(q), torch.randn_like(qd)
    b1 = eng.backward(ck, a, None, dt, S, mm, g1q, g1d)
    b2 = eng.backward(ck, a, None, dt, S, mm, g2q, g2d)
    b3 = eng.backward(ck, a, None, dt, S, mm, g1q + 2 * g2q, g1d + 2 * g2d)
    for x1, x2, x3 in zip(b1[:3], b2[:3], b3[:3]):
        assert torch.isfinite(x3).all()
        assert (x3 - (x1 + 2 * x2)).abs().max() <= 2e-4 * x3.abs().max()
    b0 = eng.backward(ck, a, None, dt, S, mm, torch.zeros_like(q), torch.zeros_like(qd))
    assert all(float(x.abs().max()) == 0.0 for x in b0[:3])


def test_error_paths(dev):
    from diffrl_amd import capi
    t, eng = _engine("cartpole", dev)
    q = torch.zeros(2 * t.n_q, device=dev)
    qd = torch.zeros(2 * t.n_qd, device=dev)
    with pytest.raises(capi.DsimError):
        eng.forward(q, qd, torch.zeros(3, device=dev), None, 1 / 60, 4, 4, False)
    with pytest.raises(capi.DsimError):
        eng.forward(q, qd, torch.zeros_like(qd), None, 1 / 60, 0, 4, False)
    with pytest.raises(capi.DsimError):
        eng.forward(q.double(), qd, torch.zeros_like(qd), None, 1 / 60, 4, 4, False)


@pytest.mark.parametrize("env", ENVS)
def test_specialised_kernels_match_generic(env, dev, monkeypatch):
    """the per-model specialised kernel set (compile-time layout) and the generic one agree"""
    from diffrl_amd.engine import Engine
    t = template_from_golden(env)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    spec_eng = Engine(t, dev)
    assert spec_eng.variant > 0, "known model should select a specialised kernel set"
    monkeypatch.setenv("DSIM_FORCE_GENERIC", "1")
    gen_eng = Engine(t, dev)
    assert gen_eng.variant == 0
    a = _run(spec_eng, dev, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    b = _run(gen_eng, dev, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    for k in ("q", "qd", "gq", "gqd", "gact"):
        assert relerr(a[k], b[k]) < 1e-5, k


@pytest.mark.parametrize("env", ENVS)
def test_first_substep_intermediates_vs_reference(env, dev):
    """Not only boundary tensors: the forward intermediates of the first substep (X_sc, S, v, a, world inertias, f_tot,
    qdd), read back from the checkpoint the HIP forward kernel wrote, against the reference's recording of that substep --
    the per-phase quantities as hipcc compiled them (contraction, scheduling), not as the host harness computes them."""
    from ckpt_fields import BOUNDS, compare_with_reference, first_substep
    t, eng = _engine(env, dev)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    r = _run(eng, dev, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    err = compare_with_reference(t, first_substep(t, r["ckpt"]), g, relerr)
    for k, e in err.items():
        assert e < BOUNDS[k], (k, e)


@pytest.mark.parametrize("env", ["ant", "humanoid", "hopper", "cheetah"])
def test_helper_wave_kernels_match_single_wave(env, dev, monkeypatch):
    """Models with ground contacts run with a helper wavefront per environment while all environments of a launch are
    resident (contacts / contacts^T / per-dof cotangents next to the main wave's blocks); beyond that, and under
    DSIM_HELPER=0, the single-wave kernels run.  Same arithmetic in the same order: bit-identical results, operator level
    and through the fused environment kernels."""
    from diffrl_amd import envs as E
    from diffrl_amd.engine import Engine
    t = template_from_golden(env)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSIM_HELPER", mode)
        eng = Engine(t, dev)
        out[mode] = _run(eng, dev, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), dt, S, mm, g["gq_out"], g["gqd_out"])
    for k in ("q", "qd", "gq", "gqd", "gact", "ckpt"):
        assert np.array_equal(out["1"][k], out["0"][k]), k
    cls = {"ant": E.AntEnv, "humanoid": E.HumanoidEnv, "hopper": E.HopperEnv, "cheetah": E.CheetahEnv}[env]
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSIM_HELPER", mode)
        e = cls(num_envs=64, device="cuda:0", no_grad=False, stochastic_init=False)
        e.initialize_trajectory()
        torch.manual_seed(3)
        a = torch.randn((4, 64, e.num_actions), device=dev).tanh().requires_grad_(True)
        tot = 0.0
        for s in range(4):
            obs, rew, done, info = e.step(a[s])
            tot = tot - rew.sum() + 0.01 * obs.sum()
        tot.backward()
        res[mode] = (obs.detach().cpu().numpy(), rew.detach().cpu().numpy(), a.grad.cpu().numpy())
    for x, y in zip(res["1"], res["0"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("env", ["ant", "hopper", "cheetah", "cartpole"])
def test_pair_kernels_match_one_environment_per_wave(env, dev, monkeypatch):
    """Launches beyond the helper-wave capacity run the FORWARD of the models that fit 32 lanes with two environments per
    wavefront (dsim_hip.hip: DSIM_MODE_PAIR; DSIM_PAIR=0 / 1 forces the choice).  Bit-identical per environment to the
    one-environment kernels -- state, observations, rewards, and the gradients the (always one-environment) adjoint computes
    from the checkpoint the pair kernel wrote -- with an odd number of environments (the last wave carries one)."""
    from diffrl_amd import envs as E
    from diffrl_amd.engine import Engine
    t = template_from_golden(env)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    monkeypatch.setenv("DSIM_HELPER", "0")
    n = 67
    reps = n // g["q_in"].shape[0] + 1
    tile = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:n])
    rng = np.random.default_rng(5)
    q, qd, act = tile(g["q_in"]), tile(g["qd_in"]), tile(g["act_in"])
    act = (act + 0.1 * rng.normal(size=act.shape)).astype(np.float32)     # (every environment its own trajectory)
    gq, gqd = rng.normal(size=q.shape).astype(np.float32), rng.normal(size=qd.shape).astype(np.float32)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSIM_PAIR", mode)
        eng = Engine(t, dev)
        out[mode] = _run(eng, dev, q, qd, act, None, dt, S, mm, gq, gqd)
    for k in ("q", "qd", "gq", "gqd", "gact"):
        assert np.isfinite(out["1"][k]).all() and np.array_equal(out["1"][k], out["0"][k]), k
    r = out["1"]
    assert relerr(r["q"][:g["q_in"].shape[0]], out["0"]["q"][:g["q_in"].shape[0]]) == 0
    cls = {"ant": E.AntEnv, "hopper": E.HopperEnv, "cheetah": E.CheetahEnv, "cartpole": E.CartPoleSwingUpEnv}[env]
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSIM_PAIR", mode)
        torch.manual_seed(12)
        e = cls(num_envs=65, device="cuda:0", no_grad=False, stochastic_init=True, seed=7)
        e.reset()
        e.initialize_trajectory()
        torch.manual_seed(3)
        a = torch.randn((4, 65, e.num_actions), device=dev).tanh().requires_grad_(True)
        tot = 0.0
        for s in range(4):
            obs, rew, done, info = e.step(a[s])
            tot = tot - rew.sum() + 0.01 * obs.sum()
        tot.backward()
        res[mode] = (obs.detach().cpu().numpy(), rew.detach().cpu().numpy(), a.grad.cpu().numpy())
    for x, y in zip(res["1"], res["0"]):
        assert np.isfinite(x).all() and np.array_equal(x, y)
    # episodes that end inside the window (3 steps): restarts from the noisy start-state pool, obs_before_reset, done flags -- the
    # two environments of a wave take different branches of the bookkeeping
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSIM_PAIR", mode)
        torch.manual_seed(12)    # (the host-side reset() draws its start states from torch's global generator, like the reference)
        e = cls(num_envs=65, device="cuda:0", no_grad=False, stochastic_init=True, seed=11, episode_length=3)
        e.reset()
        e.progress_buf[::2] += 1     # (half of the environments one step ahead: the halves of a wave restart at different steps)
        e.initialize_trajectory()
        torch.manual_seed(4)
        a = torch.randn((7, 65, e.num_actions), device=dev).tanh().requires_grad_(True)
        tot, dones, before = 0.0, [], []
        for s in range(7):
            obs, rew, done, info = e.step(a[s])
            tot = tot - rew.sum() + 0.01 * obs.sum() + 0.02 * info["obs_before_reset"].sum()
            dones.append(done.detach().cpu().numpy().copy())
            before.append(info["obs_before_reset"].detach().cpu().numpy())
        tot.backward()
        res[mode] = (obs.detach().cpu().numpy(), rew.detach().cpu().numpy(), np.stack(dones), np.stack(before), a.grad.cpu().numpy(),
                     e.progress_buf.cpu().numpy())
    assert res["1"][2].any() and not res["1"][2].all()
    for x, y in zip(res["1"], res["0"]):
        assert np.isfinite(x).all() and np.array_equal(x, y)


@pytest.mark.parametrize("env", ["ant", "humanoid", "snu"])
def test_lean_checkpoint_mode(env, dev):
    """DSIM_CKPT_LEAN (include/dsim.h): rows of (q, qd) only, the adjoint launch recomputes the forward phases -- identical
    gradients (bit for bit: same code on the same inputs), a fraction of the checkpoint memory"""
    from diffrl_amd.engine import Engine
    t = template_from_golden(env)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    mact = g.get("muscle_act_in")
    res, words = {}, {}
    for mode in ("full", "lean"):
        eng = Engine(t, dev, ckpt_mode=mode)
        res[mode] = _run(eng, dev, g["q_in"], g["qd_in"], g["act_in"], mact, dt, S, mm, g["gq_out"], g["gqd_out"])
        words[mode] = int(eng._lib.dsim_ckpt_floats_mm(eng._h, S, mm))
    for k in res["full"]:
        if k != "ckpt":
            assert np.array_equal(res["full"][k], res["lean"][k]), k
    assert res["lean"]["ckpt"].shape[1] == words["lean"] < 0.2 * words["full"]


@pytest.mark.parametrize("env", ["ant", "snu"])
def test_literal_backward_in_the_lean_checkpoint_mode(env, dev):
    """the literal call reads (q, qd) from the head of the first substep's row and the first group's inverse: both checkpoint modes
    have them at their own row strides -- same result (the adjoint's part bit for bit, the radial part from the same inputs)"""
    from diffrl_amd.engine import Engine
    t = template_from_golden(env)
    g = golden(env + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    T = lambda a: torch.tensor(a, device=dev).reshape(-1)  # noqa: E731
    mact = T(g["muscle_act_in"]) if "muscle_act_in" in g else None
    out = {}
    for mode in ("full", "lean"):
        eng = Engine(t, dev, ckpt_mode=mode)
        qo, qdo, ck = eng.forward(T(g["q_in"]), T(g["qd_in"]), T(g["act_in"]), mact, dt, S, mm, True)
        out[mode] = eng.backward(ck, T(g["act_in"]), mact, dt, S, mm, T(g["gq_out"]), T(g["gqd_out"]), literal=True)[0].cpu().numpy()
    n = g["q_in"].shape[0]
    assert relerr(out["lean"].reshape(n, -1), g["gq_in"]) < 1e-3 and relerr(out["full"].reshape(n, -1), out["lean"].reshape(n, -1)) < 1e-6


def test_literal_backward_without_quaternion_joints_is_the_plain_call(dev):
    """a model without free / ball joints has no quaternion coordinates: the second launch is skipped, the outputs are the plain ones"""
    t, eng = _engine("cartpole", dev)
    g = golden("cartpole_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    T = lambda a: torch.tensor(a, device=dev).reshape(-1)  # noqa: E731
    qo, qdo, ck = eng.forward(T(g["q_in"]), T(g["qd_in"]), T(g["act_in"]), None, dt, S, mm, True)
    a = eng.backward(ck, T(g["act_in"]), None, dt, S, mm, T(g["gq_out"]), T(g["gqd_out"]))
    b = eng.backward(ck, T(g["act_in"]), None, dt, S, mm, T(g["gq_out"]), T(g["gqd_out"]), literal=True)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(a[:3], b[:3]))
