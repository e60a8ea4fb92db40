"""CPU-only: diffrl_amd's own asset loaders + ModelBuilder reproduce the reference's model constants
(tests/golden/<env>_model.npz, dumped from the reference's builder) -- from the original asset files
when they are available and from the compiled assets (.npz builder snapshots) always."""

import os

import numpy as np
import pytest

import diffrl_amd.envs.dflex_env as de
from diffrl_amd import envs
from oracle_lib import golden

CASES = [("ant", envs.AntEnv), ("humanoid", envs.HumanoidEnv), ("snu", envs.SNUHumanoidEnv),
         ("cartpole", envs.CartPoleSwingUpEnv), ("hopper", envs.HopperEnv), ("cheetah", envs.CheetahEnv)]
FIELDS = ["joint_type", "joint_parent", "joint_q_start", "joint_qd_start", "joint_X_pj", "joint_X_cm", "joint_axis",
          "body_I_m", "joint_armature", "joint_target", "joint_target_ke", "joint_target_kd", "joint_limit_lower",
          "joint_limit_upper", "joint_limit_ke", "joint_limit_kd", "contact_body", "contact_point", "contact_dist",
          "muscle_start", "muscle_links", "muscle_points", "gravity", "joint_q0"]


def _check(env, t, exact):
    g = golden(env + "_model")
    for k in FIELDS:
        a = np.asarray(getattr(t, k))
        r = np.asarray(g[k]).reshape(a.shape)
        if exact:
            assert np.array_equal(a, r), k
        else:
            assert np.allclose(a, r, rtol=1e-6, atol=1e-7), k
    mats = np.asarray(g["shape_materials"], np.float32).reshape(-1, 4)
    cm = np.asarray(g["contact_material"], np.int64)
    if cm.size:
        assert np.array_equal(t.contact_material, mats[cm])


@pytest.mark.parametrize("env,cls", CASES)
def test_compiled_asset_matches_reference_model(env, cls, monkeypatch):
    monkeypatch.setattr(de, "find_asset", lambda name: None)  # force the .npz path (what the GPU box uses)
    for mod in (envs.ant, envs.humanoid, envs.snu_humanoid, envs.cartpole_swing_up, envs.planar):
        monkeypatch.setattr(mod, "find_asset", lambda name: None)
    e = cls(num_envs=3, device="cpu", no_grad=True)
    _check(env, e.model.template(), exact=True)
    assert e.model.joint_q.numel() == 3 * e.num_joint_q


@pytest.mark.parametrize("env,cls", CASES)
def test_loader_matches_reference_model(env, cls, monkeypatch):
    # the library only looks in $DIFFRL_ASSETS / its own assets dir: the test names the reference's asset dir explicitly
    ref_assets = os.path.join(os.environ.get("DIFFRL_REFERENCE", "/root/reference"), "envs", "assets")
    if os.path.isdir(ref_assets) and not os.environ.get("DIFFRL_ASSETS"):
        monkeypatch.setenv("DIFFRL_ASSETS", ref_assets)
    probe = {"ant": "ant.xml", "humanoid": "humanoid.xml", "snu": "snu/human.xml", "cartpole": "cartpole.urdf",
             "hopper": "hopper.xml", "cheetah": "half_cheetah.xml"}[env]
    if de.find_asset(probe) is None:
        pytest.skip("original asset files not available here")
    e = cls(num_envs=2, device="cpu", no_grad=True)
    _check(env, e.model.template(), exact=True)


def test_snu_muscle_strengths():
    e = envs.SNUHumanoidEnv(num_envs=2, device="cpu", no_grad=True)
    g = golden("snu_model")
    assert np.allclose(e.muscle_strengths[:152].numpy(), g["muscle_strengths"], rtol=1e-6)


def test_step_without_gpu_fails_loudly():
    from diffrl_amd.capi import DsimError
    import torch
    e = envs.CartPoleSwingUpEnv(num_envs=2, device="cpu", no_grad=True)
    with pytest.raises(DsimError):
        e.step(torch.zeros(2, 1))
