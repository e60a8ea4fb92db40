"""CPU-only: the episode bookkeeping fused into the step (progress_buf, reset_buf, restart from the pool of start
states, obs_before_reset, the cut of the autograd graph at a restart), executed lane-serially.

Expected behaviour = what the reference does around its step with torch ops (envs/ant.py:176-234, 297-307,
humanoid.py:340-356): computed here from the PLAIN fused step (no episode handling, itself checked against reference
rollouts in test_emu_fused_env.py) plus a few lines of numpy, and from the torch environment surface for the
observation of a restarted environment."""
import numpy as np
import pytest
import torch

from emu_lib import emu_env_backward, emu_env_forward, env_spec_for, make_episode
from oracle_lib import golden, template_from_golden

DT, S, MM = 1.0 / 60.0, 16, 16
EP_LEN = 1000


def _setup(env="ant", n=5, pool=2, seed=0):
    t = template_from_golden(env)
    g = golden(env + "_rollout")
    spec, keep = env_spec_for(env, t)
    rng = np.random.default_rng(seed)
    reps = n // g["q0"].shape[0] + 1
    q = np.tile(g["q0"], (reps, 1))[:n].copy()
    qd = np.tile(g["qd0"], (reps, 1))[:n].copy()
    a = np.tile(g["actions"][0], (reps, 1))[:n].copy()
    a[0, 0] = 1.7  # clipped action: gradient must be zero there
    # env 1: episode ends by length; env 2: torso below the termination height; env 3: exploded state; env 4: both 1 and 2
    progress = np.array([3, EP_LEN - 1, 10, 20, EP_LEN - 1], np.int64)[:n].copy()
    if n >= 5:
        q[2, 1] = 0.2
        q[4, 1] = 0.2
        qd[3, 8] = 3.0e7
    pool_q = np.stack([np.tile(g["q0"][:1], (n, 1)) + 0.01 * rng.normal(size=(n, t.n_q)) for _ in range(pool)]).astype(np.float32)
    pool_q[:, :, 3:7] /= np.linalg.norm(pool_q[:, :, 3:7], axis=-1, keepdims=True)
    pool_qd = (0.1 * rng.normal(size=(pool, n, t.n_qd))).astype(np.float32)
    cnt = np.array([0, 1, 0, 5, 2], np.int32)[:n].copy()
    return t, spec, keep, q.astype(np.float32), qd.astype(np.float32), a.astype(np.float32), progress, pool_q, pool_qd, cnt


def _obs_of(env, t, q, qd):
    """observation of a freshly restarted environment (stored actions cleared) through the torch env surface"""
    from oracle_env import make_cpu_env
    e = make_cpu_env(env, q.shape[0], t)
    e.state.joint_q = torch.tensor(q).reshape(-1)
    e.state.joint_qd = torch.tensor(qd).reshape(-1)
    e.actions = torch.zeros((q.shape[0], e.num_actions))
    e.calculateObservations()
    return e.obs_buf.numpy()


@pytest.mark.parametrize("check_invalid", [True, False])
def test_forward_flags_and_restart(check_invalid):
    t, spec, keep, q, qd, a, progress, pool_q, pool_qd, cnt = _setup()
    n = q.shape[0]
    q_p, qd_p, obs_p, rew_p, _ = emu_env_forward(t, spec, q, qd, a, DT, S, MM)  # plain step
    prog, c = progress.copy(), cnt.copy()
    done = np.full(n, -7, np.int64)
    obs_before = np.full((n, spec.n_obs), np.nan, np.float32)
    ep = make_episode(prog, done, obs_before, pool_q, pool_qd, c, EP_LEN, True, check_invalid)
    q_e, qd_e, obs_e, rew_e, ck = emu_env_forward(t, spec, q, qd, a, DT, S, MM, episode=ep)

    bad = np.zeros(n, bool)
    if check_invalid:
        bad = ~(np.abs(q_p) <= 1e6).all(-1) | ~(np.abs(qd_p) <= 1e6).all(-1) | ~np.isfinite(obs_p).all(-1)
        assert bad[3] and bad.sum() == 1
    want = (obs_p[:, 0] < 0.27) | (progress + 1 > EP_LEN - 1) | bad
    assert want[[1, 2, 4]].all() and not want[0]
    np.testing.assert_array_equal(done, want.astype(np.int64))
    np.testing.assert_array_equal(prog, np.where(want, 0, progress + 1))
    np.testing.assert_array_equal(c, cnt + want)
    np.testing.assert_array_equal(obs_before, obs_p)
    np.testing.assert_array_equal(rew_e, np.where(bad, 0.0, rew_p).astype(np.float32))
    live = ~want
    np.testing.assert_array_equal(q_e[live], q_p[live])
    np.testing.assert_array_equal(qd_e[live], qd_p[live])
    np.testing.assert_array_equal(obs_e[live], obs_p[live])
    idx = np.nonzero(want)[0]
    slot = cnt[idx] % pool_q.shape[0]
    np.testing.assert_array_equal(q_e[idx], pool_q[slot, idx])
    np.testing.assert_array_equal(qd_e[idx], pool_qd[slot, idx])
    ref = _obs_of("ant", t, q_e, qd_e)
    assert np.abs(obs_e[idx] - ref[idx]).max() < 1e-5
    assert (obs_e[idx, -8:] == 0).all()  # stored actions cleared


def test_backward_cuts_the_graph_at_a_restart():
    t, spec, keep, q, qd, a, progress, pool_q, pool_qd, cnt = _setup()
    n = q.shape[0]
    q[3] = q[0]
    qd[3] = qd[0]  # no exploding state in this test (its adjoint is meaningless); env 3 stays live
    rng = np.random.default_rng(1)
    gq, gqd = rng.normal(size=q.shape).astype(np.float32), rng.normal(size=qd.shape).astype(np.float32)
    gobs = rng.normal(size=(n, spec.n_obs)).astype(np.float32)
    gobs_b = rng.normal(size=(n, spec.n_obs)).astype(np.float32)
    grew = rng.normal(size=n).astype(np.float32)
    _, _, _, _, ck_p = emu_env_forward(t, spec, q, qd, a, DT, S, MM)
    prog, c, done = progress.copy(), cnt.copy(), np.zeros(n, np.int64)
    ep = make_episode(prog, done, None, pool_q, pool_qd, c, EP_LEN, True, True)
    _, _, _, _, ck_e = emu_env_forward(t, spec, q, qd, a, DT, S, MM, episode=ep)
    d = done.astype(bool)
    assert d.any() and not d.all()
    # plain adjoint with the cotangents autograd would deliver after reset()'s in-place writes
    m = (~d)[:, None].astype(np.float32)
    want = emu_env_backward(t, spec, ck_p, a, DT, S, MM, gq * m, gqd * m, gobs * m + gobs_b, grew)
    got = emu_env_backward(t, spec, ck_e, a, DT, S, MM, gq, gqd, gobs, grew, gobs_b)
    for w, g_ in zip(want, got):
        np.testing.assert_array_equal(g_, w)
    assert got[2][0, 0] == 0.0  # clipped action
    # null cotangents == zeros
    z = emu_env_backward(t, spec, ck_e, a, DT, S, MM, np.zeros_like(gq), np.zeros_like(gqd), np.zeros_like(gobs), grew,
                         np.zeros_like(gobs))
    nul = emu_env_backward(t, spec, ck_e, a, DT, S, MM, None, None, None, grew, None)
    for w, g_ in zip(z, nul):
        np.testing.assert_array_equal(g_, w)


def test_invalid_state_drops_the_reward_cotangent():
    t, spec, keep, q, qd, a, progress, pool_q, pool_qd, cnt = _setup()
    n = q.shape[0]
    prog, c, done = progress.copy(), cnt.copy(), np.zeros(n, np.int64)
    ep = make_episode(prog, done, None, pool_q, pool_qd, c, EP_LEN, True, True)
    _, _, _, rew, ck = emu_env_forward(t, spec, q, qd, a, DT, S, MM, episode=ep)
    assert rew[3] == 0.0 and done[3] == 1
    _, _, ga = emu_env_backward(t, spec, ck, a, DT, S, MM, None, None, None, np.ones(n, np.float32), None)
    assert (ga[3] == 0).all() and np.abs(ga[0]).max() > 0


def test_multi_step_episode_rollover():
    """progress counts up across steps, rolls over at the episode length, pool entries are consumed in order"""
    t, spec, keep, q, qd, a, progress, pool_q, pool_qd, cnt = _setup(n=2, pool=3)
    q[:] = golden("ant_rollout")["q0"][:1]
    qd[:] = golden("ant_rollout")["qd0"][:1]
    prog, c, done = np.array([0, 0], np.int64), np.zeros(2, np.int32), np.zeros(2, np.int64)
    L = 3
    seen = []
    for step in range(8):
        pq = np.ascontiguousarray(pool_q[:, :2])
        pqd = np.ascontiguousarray(pool_qd[:, :2])
        ep = make_episode(prog, done, None, pq, pqd, c, L, False, False)
        q, qd, obs, rew, _ = emu_env_forward(t, spec, q, qd, 0.1 * a[:2], DT, S, MM, episode=ep)
        seen.append((int(prog[0]), int(done[0]), int(c[0])))
        if done[0]:
            np.testing.assert_array_equal(q[0], pq[(c[0] - 1) % 3, 0])
    assert seen == [(1, 0, 0), (2, 0, 0), (0, 1, 1), (1, 0, 1), (2, 0, 1), (0, 1, 2), (1, 0, 2), (2, 0, 2)]


def test_in_kernel_restart_noise_cpu_harness():
    """stochastic restarts drawn inside the step (Philox counter = (env, restart number, coordinate)): within the stated
    amplitudes around the start state, unit root rotation tilted by at most noise_angle / 2, a different draw for every
    environment and every restart, reproducible from the seed.  (The GPU test checks the distribution against reset_state().)"""
    t = template_from_golden("ant")
    g = golden("ant_rollout")
    spec, keep = env_spec_for("ant", t)
    n = 8
    q0 = np.tile(g["q0"][:1], (n, 1)).astype(np.float32)
    qd0 = np.zeros((n, t.n_qd), np.float32)
    nq = np.zeros(t.n_q, np.float32); nq[0:3] = 0.2; nq[7:] = 0.4
    nqd = np.full(t.n_qd, 0.5, np.float32)
    a = np.zeros((n, 8), np.float32)

    def run(seed, restarts):
        prog, done, cnt = np.zeros(n, np.int64), np.zeros(n, np.int64), np.zeros(n, np.int32)
        pq, pqd = q0[None].copy(), qd0[None].copy()
        out = []
        for _ in range(restarts):
            ep = make_episode(prog, done, None, pq, pqd, cnt, 1, False, False, nq, nqd, np.pi / 12.0, seed)
            qo, qdo, obs, rew, ck = emu_env_forward(t, spec, q0, qd0, a, 1.0 / 60.0, 16, 16, ep)
            assert done.sum() == n
            out.append((qo.copy(), qdo.copy()))
        assert (cnt == restarts).all()
        return out

    r = run(7, 3)
    for qo, qdo in r:
        d = qo - q0
        assert (np.abs(d[:, 0:3]) <= 0.1 + 1e-6).all() and (np.abs(d[:, 7:]) <= 0.2 + 1e-6).all()
        assert (np.abs(qdo) <= 0.25 + 1e-6).all()
        rot = qo[:, 3:7]
        assert np.allclose(np.linalg.norm(rot, axis=1), 1.0, atol=1e-6)
        cosang = np.abs((rot * q0[:, 3:7]).sum(1)).clip(max=1.0)
        assert (2.0 * np.arccos(cosang) <= np.pi / 24.0 + 1e-3).all()
        assert len(np.unique(np.round(qo, 7), axis=0)) == n          # environments differ
    assert not np.allclose(r[0][0], r[1][0]) and not np.allclose(r[1][0], r[2][0])   # restarts differ
    r2 = run(7, 3)
    assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(r, r2))
    assert not np.array_equal(run(8, 1)[0][0], r[0][0])


def test_episode_bookkeeping_with_the_pair_kernels_lane_count():
    """The forward kernels of launches beyond the helper capacity carry two environments per wavefront (dsim_hip.hip:
    DSIM_MODE_PAIR): the phase code sees 32 lanes per environment.  Episode flags, restarts from the pool, obs_before_reset and the
    checkpoint tail with that instantiation of the specialised Ant kernels: bit-identical to the 64-lane one."""
    from emu_lib import emu
    t, spec, keep, q, qd, a, progress, pool_q, pool_qd, cnt = _setup()
    n = q.shape[0]
    out = {}
    emu().dsim_emu_use_static(1)
    try:
        for half in (0, 1):
            emu().dsim_emu_set_half_wave(half)
            prog, c = progress.copy(), cnt.copy()
            done = np.full(n, -7, np.int64)
            obs_before = np.full((n, spec.n_obs), np.nan, np.float32)
            ep = make_episode(prog, done, obs_before, pool_q, pool_qd, c, EP_LEN, True, True)
            res = emu_env_forward(t, spec, q, qd, a, DT, S, MM, episode=ep)
            out[half] = tuple(res) + (prog, c, done, obs_before)
    finally:
        emu().dsim_emu_set_half_wave(0)
        emu().dsim_emu_use_static(0)
    assert out[0][7].any() and not out[0][7].all()     # some environments restarted, some did not
    for x, y in zip(out[1], out[0]):
        np.testing.assert_array_equal(x, y)
