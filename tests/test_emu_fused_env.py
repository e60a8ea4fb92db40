"""CPU-only: the fused env surface (action mapping + observation + reward + their adjoints inside the step
phases) executed lane-serially, against rollouts recorded from the REFERENCE environments."""
import numpy as np
import pytest

from emu_lib import emu_env_backward, emu_env_forward, env_spec_for
from oracle_lib import golden, relerr, template_from_golden

SUBSTEPS = {"cartpole": 4, "ant": 16, "humanoid": 48, "snu": 48, "hopper": 16, "cheetah": 16}


# <env>_rollout_mm1: MM_caching_frequency = 1, the constructor default of the reference's environment classes
# (envs/ant.py:32): mass matrix rebuilt -- and its adjoint run -- in every substep (sim.py:2113, 2475)
@pytest.mark.parametrize("env,name", [(e, e + "_rollout") for e in ["cartpole", "ant", "humanoid", "snu", "hopper", "cheetah"]]
                         + [(e, e + "_rollout_mm1") for e in ["ant", "snu", "humanoid"]])
def test_fused_rollout_vs_reference(env, name):
    t = template_from_golden(env)
    g = golden(name)
    spec, keep = env_spec_for(env, t)
    H, n = g["actions"].shape[0], g["actions"].shape[1]
    S, mm, dt = SUBSTEPS[env], int(g["mm_freq"]), 1.0 / 60.0
    q, qd = g["q0"], g["qd0"]
    tape = []
    for s in range(H):
        qo, qdo, obs, rew, ck = emu_env_forward(t, spec, q, qd, g["actions"][s], dt, S, mm)
        assert relerr(obs, g["obs"][s]) < 1e-3, s
        assert np.abs(rew - g["rew"][s]).max() < 1e-3 * max(1.0, np.abs(g["rew"]).max()), s
        tape.append((ck, qo, qdo))
        q, qd = qo, qdo
    assert relerr(q, g["q_final"]) < 1e-3
    gq, gqd = np.zeros_like(q), np.zeros_like(qd)
    ga = np.zeros_like(g["actions"])
    for s in reversed(range(H)):
        ck, qo, qdo = tape[s]
        gq, gqd, ga[s] = emu_env_backward(t, spec, ck, g["actions"][s], dt, S, mm, gq, gqd,
                                          np.zeros((n, spec.n_obs), np.float32), -np.ones(n, np.float32))
    a, r = ga.astype(np.float64), g["grad_actions"].astype(np.float64)
    assert (a * r).sum() / (np.linalg.norm(a) * np.linalg.norm(r)) > 0.9999
    tol = 1e-3
    if relerr(a, r) >= tol:
        # Is the REFERENCE-order gradient itself that sensitive here?  Perturb the start state by ~1 ulp and
        # recompute it with the scalar oracle (reference operation order): a foot contact near zero tangential
        # velocity makes the smooth-Coulomb term ill-conditioned ("gradients are numerically unstable around
        # |vt| = 0", dflex/dflex/sim.py:1200).  Measured on the SNU golden: 2.7e-3 .. 5.5e-3 for env 0.
        from oracle_env import rollout_grad
        rng = np.random.default_rng(0)
        q0p = (g["q0"].astype(np.float64) * (1.0 + 1e-7 * rng.normal(size=g["q0"].shape))).astype(np.float32)
        _, _, gp = rollout_grad(env, t, q0p, g["qd0"], g["actions"])
        tol = max(tol, 3.0 * relerr(gp, r))
    assert relerr(a, r) < tol


def test_fused_obs_cotangent_matches_finite_difference():
    """d(sum w.obs + rew)/d(actions) through ONE fused Ant step vs central differences of the fused forward"""
    t = template_from_golden("ant")
    g = golden("ant_rollout")
    spec, keep = env_spec_for("ant", t)
    rng = np.random.default_rng(0)
    q, qd, a = g["q0"][:1], g["qd0"][:1], (0.5 * g["actions"][0][:1]).astype(np.float32)
    w = rng.normal(0, 1, (1, 37)).astype(np.float32)
    qo, qdo, obs, rew, ck = emu_env_forward(t, spec, q, qd, a, 1 / 60, 16, 16)
    _, _, ga = emu_env_backward(t, spec, ck, a, 1 / 60, 16, 16, np.zeros_like(q), np.zeros_like(qd), w,
                                np.ones(1, np.float32))

    def f(x):
        _, _, o, r, _ = emu_env_forward(t, spec, q, qd, x, 1 / 60, 16, 16)
        return float((o.astype(np.float64) * w).sum() + r.astype(np.float64).sum())
    fd = np.zeros(8)
    for k in range(8):
        e = np.zeros_like(a)
        e[0, k] = 5e-3
        fd[k] = (f(a + e) - f(a - e)) / 1e-2
    assert np.abs(ga[0] - fd).max() < 2e-2 * max(1.0, np.abs(fd).max())


def test_sanitize_grads_scrubs_the_returned_cotangents_only_when_asked():
    """dsim_env_spec.sanitize_grads (include/dsim.h): the adjoint launch writes 0 for every non-finite cotangent it returns --
    torch.nan_to_num(grad, 0, 0, 0) of the reference's per-step hooks on joint_q / joint_qd / actions (envs/humanoid.py:195-206).
    A NaN / inf fed into one environment's observation cotangent comes out as zeros there with the flag, as non-finite values
    without it; the other environment is untouched either way (bit-identical)."""
    from emu_lib import emu_env_backward, emu_env_forward, env_spec_for
    from oracle_lib import golden, template_from_golden
    t = template_from_golden("humanoid")
    g = golden("humanoid_rollout")
    spec, keep = env_spec_for("humanoid", t)
    n = 2
    q, qd, a = g["q0"][:n], g["qd0"][:n], g["actions"][0][:n]
    f = emu_env_forward(t, spec, q, qd, a, 1 / 60, 48, 48)
    rng = np.random.default_rng(5)
    gq, gqd = rng.normal(size=q.shape).astype(np.float32), rng.normal(size=qd.shape).astype(np.float32)
    gobs, grew = rng.normal(size=(n, spec.n_obs)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    clean = emu_env_backward(t, spec, f[4], a, 1 / 60, 48, 48, gq, gqd, gobs, grew)
    bad = gobs.copy()
    bad[0, 3] = np.nan
    bad[0, 20] = np.inf
    out = {}
    for flag in (0, 1):
        spec.sanitize_grads = flag
        out[flag] = emu_env_backward(t, spec, f[4], a, 1 / 60, 48, 48, gq, gqd, bad, grew)
    spec.sanitize_grads = 0
    assert not all(np.isfinite(x[0]).all() for x in out[0]), "without the flag the non-finite cotangent must come through"
    for x, y, c in zip(out[1], out[0], clean):
        assert np.isfinite(x).all()
        np.testing.assert_array_equal(x[1], c[1])                       # the healthy environment: untouched
        np.testing.assert_array_equal(x[0][np.isfinite(y[0])], y[0][np.isfinite(y[0])])   # finite entries pass through unchanged
        assert (x[0][~np.isfinite(y[0])] == 0.0).all()                   # non-finite ones become exactly 0
