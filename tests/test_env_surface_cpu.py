"""CPU-only: diffrl_amd's torch env surface (protocol, action mapping, observations, rewards) chained with
the scalar oracle reproduces the rollouts recorded from the REFERENCE environments."""
import numpy as np
import pytest

from oracle_env import rollout_grad
from oracle_lib import golden, relerr, template_from_golden


@pytest.mark.parametrize("env", ["cartpole", "ant", "humanoid", "snu", "hopper", "cheetah"])
def test_torch_env_surface_plus_oracle_vs_reference_rollout(env):
    t = template_from_golden(env)
    g = golden(env + "_rollout")
    obs, rew, ga = rollout_grad(env, t, g["q0"], g["qd0"], g["actions"])
    assert relerr(obs, g["obs"]) < 1e-5
    assert np.abs(rew - g["rew"]).max() < 1e-5 * max(1.0, np.abs(g["rew"]).max())
    assert relerr(ga, g["grad_actions"]) < 2e-4
