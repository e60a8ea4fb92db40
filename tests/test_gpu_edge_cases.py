"""-m gpu: edge cases through the real kernels -- user-built articulations (no specialised kernel set: the
generic runtime-layout kernels run), N = 1, a large non-power-of-two N, actions outside the clip range."""
import numpy as np
import pytest
import torch

from oracle_lib import golden, oracle_backward, project_tangent, relerr, step_grad_tolerance, template_from_golden
from test_edge_cases_cpu import _chain, _random_tree, _tree_states

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(eng, t, q, qd, act, dt, S, mm, gq, gqd):
    dev = torch.device(DEV)
    T = lambda a: torch.tensor(a, device=dev).reshape(-1)  # noqa: E731
    qo, qdo, ck = eng.forward(T(q), T(qd), T(act), None, dt, S, mm, True)
    r = eng.backward(ck, T(act), None, dt, S, mm, T(gq), T(gqd))
    torch.cuda.synchronize()
    n = q.shape[0]
    return (qo.cpu().numpy().reshape(n, -1), qdo.cpu().numpy().reshape(n, -1), r[0].cpu().numpy().reshape(n, -1),
            r[1].cpu().numpy().reshape(n, -1), r[2].cpu().numpy().reshape(n, -1))


@pytest.mark.parametrize("n_links,shapes,floating", [(1, True, False), (4, True, True), (7, True, False)])
def test_generic_kernels_on_user_models(n_links, shapes, floating):
    from diffrl_amd.engine import Engine
    t = _chain(n_links, shapes, floating)
    eng = Engine(t, DEV)
    assert eng.variant == 0, "a user-built model must run on the generic (runtime layout) kernels"
    rng = np.random.default_rng(n_links)
    n = 33
    q = np.tile(t.joint_q0, (n, 1)) + rng.normal(0, 0.2, (n, t.n_q)).astype(np.float32)
    if floating:
        q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    qd = rng.normal(0, 0.5, (n, t.n_qd)).astype(np.float32)
    act = rng.normal(0, 1.0, (n, t.n_qd)).astype(np.float32)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    for S, mm in [(1, 1), (6, 4)]:
        dt = S / 960.0
        o = oracle_backward(t, q, qd, act, None, dt, S, mm, gq, gqd)
        qo, qdo, g_q, g_qd, g_a = _run(eng, t, q, qd, act, dt, S, mm, gq, gqd)
        assert relerr(qo, o["q_out"]) < 1e-4 and relerr(qdo, o["qd_out"]) < 1e-3
        assert relerr(project_tangent(t, q, g_q), project_tangent(t, q, o["gq"])) < 1e-3
        assert relerr(g_qd, o["gqd"]) < 1e-3 and relerr(g_a, o["gact"]) < 1e-3


@pytest.mark.parametrize("seed,floating", [(0, True), (1, False), (2, True), (3, True), (4, False)])
def test_generic_kernels_on_random_trees(seed, floating):
    """branching trees numbered breadth-first (CSR-list code paths), revolute / prismatic / ball joints, mixed shapes"""
    from diffrl_amd.engine import Engine
    t, parents = _random_tree(seed, floating)
    eng = Engine(t, DEV)
    assert eng.variant == 0
    rng = np.random.default_rng(100 + seed)
    q, qd, act = _tree_states(t, rng, 17)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    for S, mm in [(4, 2), (3, 3)]:
        dt = S / 960.0
        o = oracle_backward(t, q, qd, act, None, dt, S, mm, gq, gqd)
        qo, qdo, g_q, g_qd, g_a = _run(eng, t, q, qd, act, dt, S, mm, gq, gqd)
        assert relerr(qo, o["q_out"]) < 1e-4 and relerr(qdo, o["qd_out"]) < 1e-3
        err = dict(gq=relerr(project_tangent(t, q, g_q), project_tangent(t, q, o["gq"])), gqd=relerr(g_qd, o["gqd"]),
                   gact=relerr(g_a, o["gact"]))
        tol = step_grad_tolerance(t, q, qd, act, None, dt, S, mm, gq, gqd, err, ref=o)   # 1e-3, or probed
        assert all(err[k] < tol[k] for k in err), (err, tol)


# (7, True, 40): bodies with several chunks of muscle rows, the last one of each filled up with rows nothing writes (dsim_layout.hpp)
@pytest.mark.parametrize("seed,floating,muscles", [(5, True, 4), (6, False, 4), (7, True, 40)])
def test_generic_kernels_on_random_trees_with_muscles(seed, floating, muscles):
    from diffrl_amd.engine import Engine
    t, parents = _random_tree(seed, floating, muscles=muscles)
    eng = Engine(t, DEV)
    dev = torch.device(DEV)
    rng = np.random.default_rng(200 + seed)
    n = 9
    q, qd, act = _tree_states(t, rng, n)
    mact = rng.uniform(0.0, 30.0 if muscles <= 4 else 10.0, (n, t.n_muscles)).astype(np.float32)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    S, mm = 4, 2
    dt = S / 960.0
    o = oracle_backward(t, q, qd, act, mact, dt, S, mm, gq, gqd)
    T = lambda a: torch.tensor(a, device=dev).reshape(-1)  # noqa: E731
    qo, qdo, ck = eng.forward(T(q), T(qd), T(act), T(mact), dt, S, mm, True)
    r = eng.backward(ck, T(act), T(mact), dt, S, mm, T(gq), T(gqd))
    torch.cuda.synchronize()
    N = lambda x: x.cpu().numpy().reshape(n, -1)  # noqa: E731
    assert relerr(N(qo), o["q_out"]) < 1e-4 and relerr(N(qdo), o["qd_out"]) < 1e-3
    err = dict(gq=relerr(project_tangent(t, q, N(r[0])), project_tangent(t, q, o["gq"])), gqd=relerr(N(r[1]), o["gqd"]),
               gact=relerr(N(r[2]), o["gact"]), gmact=relerr(N(r[3]), o["gmact"]))
    tol = step_grad_tolerance(t, q, qd, act, mact, dt, S, mm, gq, gqd, err, ref=o)   # 1e-3, or probed
    assert all(err[k] < tol[k] for k in err), (err, tol)


@pytest.mark.parametrize("n", [1, 3, 100003])
def test_env_count_extremes(n):
    """one env, a prime-ish count and ~1e5 envs: every env of a replicated batch gives the same answer"""
    from diffrl_amd.engine import Engine
    t = template_from_golden("ant")
    g = golden("ant_step")
    eng = Engine(t, DEV)
    dev = torch.device(DEV)
    S, mm, dt = 16, 16, 1 / 60
    k = 9  # a state in ground contact
    q = torch.tensor(g["q_in"][k], device=dev).repeat(n)
    qd = torch.tensor(g["qd_in"][k], device=dev).repeat(n)
    a = torch.tensor(g["act_in"][k], device=dev).repeat(n)
    qo, qdo, ck = eng.forward(q, qd, a, None, dt, S, mm, True)
    gq = torch.tensor(g["gq_out"][k], device=dev).repeat(n)
    gqd = torch.tensor(g["gqd_out"][k], device=dev).repeat(n)
    r = eng.backward(ck, a, None, dt, S, mm, gq, gqd)
    torch.cuda.synchronize()
    assert relerr(qo.view(n, -1)[0].cpu().numpy(), g["q_out"][k]) < 1e-4
    assert relerr(r[2].view(n, -1)[0].cpu().numpy(), g["gact_in"][k]) < 1e-3
    for x in (qo, qdo, r[0], r[1], r[2]):
        x = x.view(n, -1)
        assert torch.equal(x[0], x[n - 1]) and torch.equal(x[0], x[n // 2])


def test_clip_range_gradients():
    """actions beyond [-1, 1] are clipped inside the kernel; their gradient is zero, the others are untouched"""
    from diffrl_amd import envs
    e = envs.AntEnv(num_envs=4, device=DEV, no_grad=False, stochastic_init=False, MM_caching_frequency=16,
                    early_termination=False)
    e.initialize_trajectory()
    a = torch.tensor([[0.5, -0.5, 1.5, -2.0, 0.0, 0.9, 1.0, -1.0]] * 4, device=DEV, requires_grad=True)
    obs, rew, done, _ = e.step(a)
    assert torch.allclose(obs[:, 29:37], torch.clip(a, -1, 1))
    (rew.sum() + obs.sum()).backward()
    g = a.grad
    assert float(g[:, 2].abs().max()) == 0.0 and float(g[:, 3].abs().max()) == 0.0
    assert float(g[:, [0, 1, 4, 5, 6, 7]].abs().min()) > 0.0


def test_half_angle_polynomials_at_and_beyond_their_range_on_the_gpu():
    """the GPU side of tests/test_edge_cases_cpu.py::test_half_angle_polynomials_at_and_beyond_their_range: revolute angles at
    the ends of the polynomial range of dsim_math.hpp::half_angle_sincos, at the switch to sinf / cosf and beyond, first-substep
    link poses (read back from the checkpoint the HIP kernel wrote) and end state against the scalar oracle"""
    from ckpt_fields import first_substep
    from diffrl_amd.engine import Engine
    from oracle_lib import oracle_forward
    t = _chain(3, True, False)
    eng = Engine(t, DEV)
    angles = np.array([0.0, 1.0, -1.0, np.pi - 1e-3, -(np.pi - 1e-3), np.pi, -np.pi, np.pi + 1e-3, -(np.pi + 1e-3), 2.5 * np.pi,
                       -7.3, 3.0, -3.1], np.float32)
    n = len(angles)
    q = np.zeros((n, t.n_q), np.float32)
    q[:, 0], q[:, 1], q[:, 2] = angles, angles[::-1], 0.5 * angles
    qd, act = np.zeros((n, t.n_qd), np.float32), np.zeros((n, t.n_qd), np.float32)
    dt, S, mm = 1.0 / 960.0, 1, 1
    T = lambda a: torch.tensor(a, device=torch.device(DEV)).reshape(-1)  # noqa: E731
    qo, qdo, ck = eng.forward(T(q), T(qd), T(act), None, dt, S, mm, True)
    torch.cuda.synchronize()
    o_q, o_qd, dbg = oracle_forward(t, q, qd, act, None, dt, S, mm, debug=True)
    assert relerr(first_substep(t, ck.cpu().numpy())["X_sc"], dbg["X_sc"]) < 1e-6
    assert relerr(qo.cpu().numpy().reshape(n, -1), o_q) < 1e-6


def test_denormal_tangential_contact_velocity_stays_finite():
    """a tangential contact velocity of ~1e-20 makes |vt|^2 denormal; v_rsq_f32 flushes it to zero (rsq = inf) and the
    correction steps would turn that into NaN for the whole environment.  dsim_inv_len_* treat a squared length below the
    smallest normal as zero length.  A floating capsule resting in the ground with such a drift: the state stays finite and
    equals the oracle's (whose exact sqrt / division handle denormals), the gradients are finite."""
    from diffrl_amd.engine import Engine
    t = _chain(1, True, True)
    eng = Engine(t, DEV)
    n = 4
    q = np.tile(t.joint_q0, (n, 1)).astype(np.float32)
    q[:, 1] = 0.03                                   # capsule radius 0.05: both end caps penetrate
    qd = np.zeros((n, t.n_qd), np.float32)
    qd[:, 3] = np.array([1e-20, -1e-20, 3e-21, 0.0], np.float32)   # world-frame linear drift along x: vt^2 = 1e-40 .. 0
    qd[:, 4] = -0.1
    act = np.zeros((n, t.n_qd), np.float32)
    gq, gqd = np.ones_like(q), np.ones_like(qd)
    dt, S, mm = 1.0 / 960.0, 1, 1
    o = oracle_backward(t, q, qd, act, None, dt, S, mm, gq, gqd)
    qo, qdo, g_q, g_qd, g_a = _run(eng, t, q, qd, act, dt, S, mm, gq, gqd)
    for x in (qo, qdo, g_q, g_qd, g_a):
        assert np.isfinite(x).all(), x
    assert relerr(qo, o["q_out"]) < 1e-5 and relerr(qdo, o["qd_out"]) < 1e-4
