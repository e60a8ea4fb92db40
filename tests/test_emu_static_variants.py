"""CPU-only: the per-model specialised code paths (compile-time layouts, `if constexpr` branches such as the bounded
one-round-trip sums) executed lane-serially and compared with the generic run-time-layout paths of the same phase code.
On the GPU the same comparison is tests/test_gpu_parity.py::test_specialised_kernels_match_generic."""
import numpy as np
import pytest

from emu_lib import emu, emu_env_backward, emu_env_forward, env_spec_for
from oracle_lib import golden, template_from_golden

SUBSTEPS = {"cartpole": 4, "ant": 16, "humanoid": 48, "snu": 48, "hopper": 16, "cheetah": 16}


@pytest.fixture
def static_mode():
    emu().dsim_emu_use_static(1)
    yield
    emu().dsim_emu_use_static(0)


@pytest.mark.parametrize("env", ["cartpole", "ant", "humanoid", "snu", "hopper", "cheetah"])
def test_specialised_paths_match_generic(env, static_mode):
    t = template_from_golden(env)
    g = golden(env + "_rollout")
    spec, keep = env_spec_for(env, t)
    S, mm, dt = SUBSTEPS[env], int(g["mm_freq"]), 1.0 / 60.0
    n = min(2, g["q0"].shape[0])
    q, qd, a = g["q0"][:n], g["qd0"][:n], g["actions"][0][:n]
    rng = np.random.default_rng(0)
    cot = [rng.normal(size=x.shape).astype(np.float32) for x in (q, qd)]
    gobs, grew = rng.normal(size=(n, spec.n_obs)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    out = {}
    for mode in (1, 0):
        emu().dsim_emu_use_static(mode)
        f = emu_env_forward(t, spec, q, qd, a, dt, S, mm)   # asserts rc == 0: a specialised variant exists for every env
        b = emu_env_backward(t, spec, f[4], a, dt, S, mm, cot[0], cot[1], gobs, grew)
        out[mode] = (f[:4], b)
    from emu_lib import reorders
    for x, y in zip(out[1][0] + out[1][1], out[0][0] + out[0][1]):
        if reorders(t):
            # trunk decomposition of the subtree / ancestor sums (dsim_trunk_sum), log-depth kinematics
            # (dsim_fwd_kinematics_scan): the same terms in a different order
            assert np.abs(x - y).max() <= 2e-5 * max(np.abs(y).max(), 1e-6)
        else:
            # same operations in the same order; the bounded sums only add exact zeros for the unused slots
            np.testing.assert_allclose(x, y, rtol=0, atol=0)


def test_unknown_model_has_no_specialised_variant(static_mode):
    """a model whose layout is not in the generated table must not silently run a specialised path"""
    import copy
    t = copy.deepcopy(template_from_golden("ant"))
    t.joint_armature = np.asarray(t.joint_armature) * 1.0
    t.contact_point = np.asarray(t.contact_point)[:-1]     # one contact less: different layout
    t.contact_dist = np.asarray(t.contact_dist)[:-1]
    t.contact_body = np.asarray(t.contact_body)[:-1]
    t.contact_material = np.asarray(t.contact_material)[:-1]
    g = golden("ant_rollout")
    spec, keep = env_spec_for("ant", t)
    with pytest.raises(AssertionError):
        emu_env_forward(t, spec, g["q0"][:1], g["qd0"][:1], g["actions"][0][:1], 1 / 60, 16, 16)
    emu().dsim_emu_use_static(0)
    emu_env_forward(t, spec, g["q0"][:1], g["qd0"][:1], g["actions"][0][:1], 1 / 60, 16, 16)


@pytest.mark.parametrize("env", ["ant", "humanoid", "snu", "cheetah"])
@pytest.mark.parametrize("static", [0, 1])
def test_four_waves_per_environment_match_one(env, static):
    """The library runs the bigger models with 4 wavefronts per environment (256 lanes: the long item lists become single
    passes).  The phase code must not depend on the lane count: same items, same arithmetic, same order per item."""
    t = template_from_golden(env)
    g = golden(env + "_rollout")
    spec, keep = env_spec_for(env, t)
    S, mm, dt = SUBSTEPS[env], int(g["mm_freq"]), 1.0 / 60.0
    n = min(2, g["q0"].shape[0])
    q, qd, a = g["q0"][:n], g["qd0"][:n], g["actions"][0][:n]
    rng = np.random.default_rng(1)
    cot = [rng.normal(size=x.shape).astype(np.float32) for x in (q, qd)]
    gobs, grew = rng.normal(size=(n, spec.n_obs)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    out = {}
    try:
        emu().dsim_emu_use_static(static)
        for waves in (4, 1):
            emu().dsim_emu_set_waves(waves)
            f = emu_env_forward(t, spec, q, qd, a, dt, S, mm)
            b = emu_env_backward(t, spec, f[4], a, dt, S, mm, cot[0], cot[1], gobs, grew)
            out[waves] = (f[:4], b)
    finally:
        emu().dsim_emu_set_waves(1)
        emu().dsim_emu_use_static(0)
    from emu_lib import reorders
    reord = static and reorders(t)   # the one-wave specialised kernels of a deep tree: trunk order, log-depth kinematics
    for x, y in zip(out[4][0] + out[4][1], out[1][0] + out[1][1]):
        if reord:
            assert np.abs(x - y).max() <= 2e-5 * max(np.abs(y).max(), 1e-6)
        else:
            np.testing.assert_allclose(x, y, rtol=0, atol=0)


@pytest.mark.parametrize("env", ["ant", "humanoid", "snu"])
def test_lean_checkpoint_mode_gives_identical_gradients(env):
    """DSIM_CKPT_LEAN: the checkpoint row of a substep holds only (q, qd) and the adjoint recomputes the forward phases
    with the forward pass's own code: same inputs, same operations -> bit-identical gradients, a fraction of the memory."""
    from emu_lib import ckpt_floats
    t = template_from_golden(env)
    g = golden(env + "_rollout")
    spec, keep = env_spec_for(env, t)
    S, mm, dt = SUBSTEPS[env], int(g["mm_freq"]), 1.0 / 60.0
    n = min(2, g["q0"].shape[0])
    q, qd, a = g["q0"][:n], g["qd0"][:n], g["actions"][0][:n]
    rng = np.random.default_rng(2)
    cot = [rng.normal(size=x.shape).astype(np.float32) for x in (q, qd)]
    gobs, grew = rng.normal(size=(n, spec.n_obs)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    out, words = {}, {}
    try:
        for lean in (1, 0):
            emu().dsim_emu_set_ckpt_lean(lean)
            words[lean] = ckpt_floats(t, S, mm)
            f = emu_env_forward(t, spec, q, qd, a, dt, S, mm)
            assert f[4].shape[1] == words[lean]
            b = emu_env_backward(t, spec, f[4], a, dt, S, mm, cot[0], cot[1], gobs, grew)
            out[lean] = (f[:4], b)
    finally:
        emu().dsim_emu_set_ckpt_lean(0)
    for x, y in zip(out[1][0] + out[1][1], out[0][0] + out[0][1]):
        np.testing.assert_allclose(x, y, rtol=0, atol=0)
    assert words[1] < 0.2 * words[0]


@pytest.mark.parametrize("env", ["cartpole", "ant", "hopper", "cheetah"])
def test_pair_kernels_lane_count_is_bit_identical(env, static_mode):
    """Two environments per wavefront (dsim_hip.hip: DSIM_MODE_PAIR): the phase code then sees 32 lanes per environment, and
    every variant choice that depends on Exec::NL is made anew.  The forward -- the direction that ships pair kernels -- must
    give the bits of the 64-lane kernels (state, observations, reward, checkpoint); the adjoint phases, instantiated for A/B
    builds only, may re-order sums."""
    t = template_from_golden(env)
    g = golden(env + "_rollout")
    spec, keep = env_spec_for(env, t)
    S, mm, dt = SUBSTEPS[env], int(g["mm_freq"]), 1.0 / 60.0
    n = min(2, g["q0"].shape[0])
    q, qd, a = g["q0"][:n], g["qd0"][:n], g["actions"][0][:n]
    rng = np.random.default_rng(0)
    cot = [rng.normal(size=x.shape).astype(np.float32) for x in (q, qd)]
    gobs, grew = rng.normal(size=(n, spec.n_obs)).astype(np.float32), rng.normal(size=n).astype(np.float32)
    out = {}
    try:
        for half in (0, 1):
            emu().dsim_emu_set_half_wave(half)
            f = emu_env_forward(t, spec, q, qd, a, dt, S, mm)
            b = emu_env_backward(t, spec, f[4], a, dt, S, mm, cot[0], cot[1], gobs, grew)
            out[half] = (f, b)
    finally:
        emu().dsim_emu_set_half_wave(0)
    for x, y in zip(out[1][0], out[0][0]):
        np.testing.assert_array_equal(x, y)
    for x, y in zip(out[1][1], out[0][1]):
        assert np.abs(x - y).max() <= 2e-5 * max(np.abs(y).max(), 1e-6)


def test_models_beyond_32_lanes_have_no_pair_kernels(static_mode):
    t = template_from_golden("humanoid")
    g = golden("humanoid_rollout")
    spec, keep = env_spec_for("humanoid", t)
    emu().dsim_emu_set_half_wave(1)
    try:
        with pytest.raises(AssertionError):
            emu_env_forward(t, spec, g["q0"][:1], g["qd0"][:1], g["actions"][0][:1], 1 / 60, 48, 48)
    finally:
        emu().dsim_emu_set_half_wave(0)
