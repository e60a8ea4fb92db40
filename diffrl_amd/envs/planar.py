"""Planar-root locomotion environments: Hopper and HalfCheetah (reference: envs/hopper.py, envs/cheetah.py).
The MJCF root body carries two slide joints and one hinge, so the models have no free joint: the first three
coordinates are (x, z-height, pitch); observations are [q[1:], qd]."""
import math
import os

import numpy as np
import torch

from .. import dflex as df
from ..utils import load_utils as lu
from ..utils import torch_utils as tu
from .dflex_env import ASSET_DIR, DFlexEnv, find_asset


class PlanarEnv(DFlexEnv):
    sim_substeps = 16
    action_strength = 200.0
    start_height = 0.0
    root_offset = (0.0, 0.0, 0.0)
    pos_noise, rot_noise, joint_noise, vel_noise = 0.05, 0.1, 0.05, (0.05, 2.0)
    asset, compiled, mjcf_kw = None, None, {}

    def _build(self):
        builder = self.make_builder()
        self.start_pos = tu.to_torch([[0.0, self.start_height]] * self.num_envs, device=self.device)
        self.start_rotation = torch.tensor([0.0], device=self.device)
        self.start_joint_q = torch.zeros(self.num_actions, device=self.device)
        self.start_joint_target = self.start_joint_q.clone()
        self._finalize(builder, ground=True)

    @classmethod
    def make_builder(cls):
        xml = find_asset(cls.asset)
        if xml is None:
            return df.sim.ModelBuilder.load(os.path.join(ASSET_DIR, cls.compiled))
        b = df.sim.ModelBuilder()
        lu.parse_mjcf(xml, b, **cls.mjcf_kw)
        # the root joint frame: MJCF is z-up, the simulator y-up
        b.joint_X_pj[0] = df.transform(cls.root_offset, df.quat_from_axis_angle((1.0, 0.0, 0.0), -math.pi * 0.5))
        return b

    def apply_actions(self, actions):
        self.actions = actions.clone()
        self.state.joint_act.view(self.num_envs, -1)[:, 3:] = actions * self.action_strength

    def reset_state(self, env_ids):
        q, qd = self._q(), self._qd()
        k, dev = len(env_ids), self.device
        q[env_ids, 0:2] = self.start_pos[env_ids, :].clone()
        q[env_ids, 2] = self.start_rotation.clone()
        q[env_ids, 3:] = self.start_joint_q.clone()
        qd[env_ids, :] = 0.0
        if self.stochastic_init:
            q[env_ids, 0:2] = q[env_ids, 0:2] + self.pos_noise * (torch.rand(size=(k, 2), device=dev) - 0.5) * 2.0
            q[env_ids, 2] = (torch.rand(k, device=dev) - 0.5) * self.rot_noise
            q[env_ids, 3:] = q[env_ids, 3:] + self.joint_noise * (torch.rand(size=(k, self.num_joint_q - 3), device=dev) - 0.5) * 2.0
            qd[env_ids, :] = self.vel_noise[0] * (torch.rand(size=(k, self.num_joint_qd), device=dev) - 0.5) * self.vel_noise[1]
        self.actions = self.actions.clone()
        self.actions[env_ids, :] = 0.0

    def _deterministic_start_state(self):
        q, qd = super()._deterministic_start_state()
        if self.stochastic_init:
            # a stochastic reset_state() REPLACES the root angle by its noise term (envs/hopper.py:195, cheetah.py:183), it
            # does not add it to start_rotation: the start state the in-kernel noise is added to has a zero root angle
            q[..., 2] = 0.0
        return q, qd

    def reset_noise(self):
        nq = np.empty(self.num_joint_q, np.float32)
        nq[0:2] = self.pos_noise * 2.0
        nq[2] = self.rot_noise
        nq[3:] = self.joint_noise * 2.0
        return nq, np.full(self.num_joint_qd, self.vel_noise[0] * self.vel_noise[1], np.float32), 0.0

    def calculateObservations(self):
        self.obs_buf = torch.cat([self._q()[:, 1:], self._qd()], dim=-1)

    def _planar_spec(self, rew_kind, **kw):
        from .. import capi
        self._act_scale_dev = torch.full((self.num_actions,), self.action_strength, device=self.device)
        return capi.make_env_spec(capi.ENV_PLANAR, rew_kind, self.num_actions, self.num_observations,
                                  self._act_scale_dev.data_ptr(), act_offset=3, action_penalty=self.action_penalty, **kw)


class HopperEnv(PlanarEnv):
    asset, compiled = "hopper.xml", "hopper.npz"
    mjcf_kw = dict(density=1000.0, stiffness=0.0, damping=2.0, contact_ke=2.e+4, contact_kd=1.e+3, contact_kf=1.e+3,
                   contact_mu=0.9, limit_ke=1.e+3, limit_kd=1.e+1, armature=1.0, radians=True, load_stiffness=True)
    termination_height = -0.45
    termination_angle = np.pi / 6.0
    termination_height_tolerance = 0.15
    termination_angle_tolerance = 0.05
    height_rew_scale = 1.0
    action_penalty = -1e-1

    def __init__(self, render=False, device="cuda:0", num_envs=4096, seed=0, episode_length=1000, no_grad=True,
                 stochastic_init=False, MM_caching_frequency=1, early_termination=True):
        super().__init__(num_envs, 11, 3, episode_length, MM_caching_frequency, seed, no_grad, render, device)
        self.stochastic_init = stochastic_init
        self.early_termination = early_termination
        self.height_terminate = bool(early_termination)
        self._build()

    def fused_spec(self):
        from .. import capi
        return self._planar_spec(capi.REW_HOPPER, termination_height=self.termination_height,
                                 termination_tolerance=self.termination_height_tolerance,
                                 height_rew_scale=self.height_rew_scale,
                                 cartpole_penalties=(self.termination_angle, 0.0, 0.0, 0.0))

    def _may_reset(self):
        return self.early_termination or getattr(self, "_progress_hi", 0) > self.episode_length - 1

    def flag_resets(self):
        super().flag_resets()
        if self.early_termination:
            self.reset_buf = torch.where(self.obs_buf[:, 0] < self.termination_height, torch.ones_like(self.reset_buf),
                                         self.reset_buf)

    def calculateReward(self):
        o = self.obs_buf
        hr = torch.clip(o[:, 0] - (self.termination_height + self.termination_height_tolerance), -1.0, 0.3)
        hr = torch.where(hr < 0.0, -200.0 * hr * hr, hr)
        hr = torch.where(hr > 0.0, self.height_rew_scale * hr, hr)
        angle_reward = 1.0 * (-o[:, 1] ** 2 / (self.termination_angle ** 2) + 1.0)
        self.rew_buf = o[:, 5] + hr + angle_reward + torch.sum(self.actions ** 2, dim=-1) * self.action_penalty
        self.flag_resets()


class CheetahEnv(PlanarEnv):
    asset, compiled = "half_cheetah.xml", "half_cheetah.npz"
    mjcf_kw = dict(density=1000.0, stiffness=0.0, damping=1.0, contact_ke=2.e+4, contact_kd=1.e+3, contact_kf=1.e+3,
                   contact_mu=1.0, limit_ke=1.e+3, limit_kd=1.e+1, armature=0.1, radians=True, load_stiffness=True)
    start_height = -0.2
    root_offset = (0.0, 1.0, 0.0)
    pos_noise, rot_noise, joint_noise, vel_noise = 0.1, 0.2, 0.1, (0.5, 1.0)
    action_penalty = -0.1

    def __init__(self, render=False, device="cuda:0", num_envs=4096, seed=0, episode_length=1000, no_grad=True,
                 stochastic_init=False, MM_caching_frequency=1, early_termination=False):
        super().__init__(num_envs, 17, 6, episode_length, MM_caching_frequency, seed, no_grad, render, device)
        self.stochastic_init = stochastic_init
        self.early_termination = early_termination
        self._build()

    def fused_spec(self):
        from .. import capi
        return self._planar_spec(capi.REW_CHEETAH)

    def _may_reset(self):
        return getattr(self, "_progress_hi", 0) > self.episode_length - 1

    def calculateReward(self):
        self.rew_buf = self.obs_buf[:, 8] + torch.sum(self.actions ** 2, dim=-1) * self.action_penalty
        self.flag_resets()
