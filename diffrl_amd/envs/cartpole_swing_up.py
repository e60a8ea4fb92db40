"""CartPoleSwingUp (reference: envs/cartpole_swing_up.py).  Fixed base + prismatic cart + revolute
pole, no ground, 4 substeps."""
import math
import os

import numpy as np
import torch

from .. import dflex as df
from ..utils import load_utils as lu
from ..utils import torch_utils as tu
from .dflex_env import ASSET_DIR, DFlexEnv, find_asset


class CartPoleSwingUpEnv(DFlexEnv):
    sim_substeps = 4
    keep_act_on_clear = True
    action_strength = 1000.0
    pole_angle_penalty = 1.0
    pole_velocity_penalty = 0.1
    cart_position_penalty = 0.05
    cart_velocity_penalty = 0.1
    cart_action_penalty = 0.0

    def __init__(self, render=False, device="cuda:0", num_envs=1024, seed=0, episode_length=240, no_grad=True,
                 stochastic_init=False, MM_caching_frequency=1, early_termination=False):
        super().__init__(num_envs, 5, 1, episode_length, MM_caching_frequency, seed, no_grad, render, device)
        self.stochastic_init = stochastic_init
        self.early_termination = early_termination
        builder = self.make_builder()
        builder.joint_q[1] = -math.pi
        self._finalize(builder, ground=False)
        self.start_joint_q = self.state.joint_q.clone()
        self.start_joint_qd = self.state.joint_qd.clone()

    @staticmethod
    def make_builder():
        urdf = find_asset("cartpole.urdf")
        if urdf is None:
            return df.sim.ModelBuilder.load(os.path.join(ASSET_DIR, "cartpole.npz"))
        b = df.sim.ModelBuilder()
        base = df.transform((0.0, 2.5, 0.0), df.quat_from_axis_angle((1.0, 0.0, 0.0), -math.pi * 0.5))
        lu.urdf_load(b, urdf, base, floating=False, shape_kd=1e4, limit_kd=1.0)
        return b

    def fused_spec(self):
        from .. import capi
        self._act_scale_dev = torch.full((1,), self.action_strength, device=self.device)
        return capi.make_env_spec(capi.ENV_CARTPOLE, capi.REW_CARTPOLE, 1, 5, self._act_scale_dev.data_ptr(),
                                  action_penalty=self.cart_action_penalty,
                                  cartpole_penalties=(self.pole_angle_penalty, self.pole_velocity_penalty,
                                                      self.cart_position_penalty, self.cart_velocity_penalty))

    def _may_reset(self):
        return getattr(self, "_progress_hi", 0) > self.episode_length - 1

    def apply_actions(self, actions):
        self.actions = actions
        self.state.joint_act.view(self.num_envs, -1)[:, 0:1] = actions * self.action_strength

    def reset_state(self, env_ids):
        q, qd = self._q(), self._qd()
        q[env_ids, :] = self.start_joint_q.view(-1, self.num_joint_q)[env_ids, :].clone()
        qd[env_ids, :] = self.start_joint_qd.view(-1, self.num_joint_qd)[env_ids, :].clone()
        if self.stochastic_init:
            k = len(env_ids)
            q[env_ids, :] = q[env_ids, :] + np.pi * (torch.rand(size=(k, self.num_joint_q), device=self.device) - 0.5)
            qd[env_ids, :] = qd[env_ids, :] + 0.5 * (torch.rand(size=(k, self.num_joint_qd), device=self.device) - 0.5)

    def reset_noise(self):
        return np.full(self.num_joint_q, np.pi, np.float32), np.full(self.num_joint_qd, 0.5, np.float32), 0.0

    def clear_grad(self, checkpoint=None):
        with torch.no_grad():
            q, qd, act = self.state.joint_q.clone(), self.state.joint_qd.clone(), self.state.joint_act.clone()
            self.state = self.model.state()
            self.state.joint_q, self.state.joint_qd, self.state.joint_act = q, qd, act

    def calculateObservations(self):
        q, qd = self._q(), self._qd()
        theta = q[:, 1:2]
        self.obs_buf = torch.cat([q[:, 0:1], qd[:, 0:1], torch.sin(theta), torch.cos(theta), qd[:, 1:2]], dim=-1)

    def calculateReward(self):
        q, qd = self._q(), self._qd()
        theta = tu.normalize_angle(q[:, 1])
        self.rew_buf = (-torch.pow(theta, 2.0) * self.pole_angle_penalty
                        - torch.pow(qd[:, 1], 2.0) * self.pole_velocity_penalty
                        - torch.pow(q[:, 0], 2.0) * self.cart_position_penalty
                        - torch.pow(qd[:, 0], 2.0) * self.cart_velocity_penalty
                        - torch.sum(self.actions ** 2, dim=-1) * self.cart_action_penalty)
        self.reset_buf = torch.where(self.progress_buf > self.episode_length - 1, torch.ones_like(self.reset_buf),
                                     self.reset_buf)
