"""Humanoid (reference: envs/humanoid.py).  22 links (one per MJCF joint), 28 coordinates, 27 dofs,
35 ground contacts, 48 substeps."""
import os

import torch

from .. import dflex as df
from ..utils import load_utils as lu
from ..utils import torch_utils as tu
from .dflex_env import ASSET_DIR, find_asset
from .locomotion import FloatingBaseEnv


class HumanoidEnv(FloatingBaseEnv):
    sim_substeps = 48
    start_height = 1.35
    target = (200.0, 0.0, 0.0)
    termination_height = 0.74
    termination_tolerance = 0.1
    height_rew_scale = 10.0
    action_penalty = -0.002
    motor_scale = 0.35
    motor_strengths = [200, 200, 200, 200, 200, 600, 400, 100, 100, 200, 200, 600, 400, 100, 100, 100, 100, 200, 100,
                       100, 200]
    sanitize_grads = True
    check_invalid = True

    def __init__(self, render=False, device="cuda:0", num_envs=4096, seed=0, episode_length=1000, no_grad=True,
                 stochastic_init=False, MM_caching_frequency=1):
        super().__init__(num_envs, 76, 21, episode_length, MM_caching_frequency, seed, no_grad, render, device)
        self.stochastic_init = stochastic_init
        self._setup_frames()
        builder = self.make_builder()
        self._place_root(builder)
        self.start_joint_q = tu.to_torch(builder.joint_q[7:], device=self.device)
        self.start_joint_target = self.start_joint_q.clone()
        self.strengths = tu.to_torch(self.motor_strengths, device=self.device).repeat((num_envs, 1))
        self._finalize(builder, ground=True)

    @staticmethod
    def make_builder():
        xml = find_asset("humanoid.xml")
        if xml is None:
            return df.sim.ModelBuilder.load(os.path.join(ASSET_DIR, "humanoid.npz"))
        b = df.sim.ModelBuilder()
        lu.parse_mjcf(xml, b, stiffness=5.0, damping=0.1, contact_ke=2.e+4, contact_kd=5.e+3, contact_kf=1.e+3,
                      contact_mu=0.75, limit_ke=1.e+3, limit_kd=1.e+1, armature=0.007, load_stiffness=True,
                      load_armature=True)
        return b

    def fused_spec(self):
        from .. import capi
        return self._locomotion_spec(capi.REW_HUMANOID, self.motor_scale * self.strengths[0],
                                     termination_tolerance=self.termination_tolerance,
                                     height_rew_scale=self.height_rew_scale, action_penalty=self.action_penalty)

    def apply_actions(self, actions):
        self.actions = actions.clone()
        self.state.joint_act.view(self.num_envs, -1)[:, 6:] = actions * self.motor_scale * self.strengths

    def calculateReward(self):
        o = self.obs_buf
        self.rew_buf = (o[:, 5] + 0.1 * o[:, 53] + o[:, 54] + self._shaped_height_reward()
                        + torch.sum(self.actions ** 2, dim=-1) * self.action_penalty)
        self._flag_resets()
