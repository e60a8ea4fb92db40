"""Shared pieces of the floating-base locomotion environments (Ant, Humanoid, SNUHumanoid): torso
observation block, reset distribution, invalid-state handling.  Reference: envs/ant.py:192-307,
envs/humanoid.py:240-367, envs/snu_humanoid.py:300-432."""
import math

import numpy as np
import torch

from .. import dflex as df
from ..utils import torch_utils as tu
from .dflex_env import DFlexEnv


class FloatingBaseEnv(DFlexEnv):
    start_height = 1.0
    start_axis_angle = ((1.0, 0.0, 0.0), -math.pi * 0.5)
    target = (10000.0, 0.0, 0.0)
    joint_vel_obs_scaling = 0.1
    randomize_joints = True      # +-0.2 rad on the joint coordinates at stochastic reset
    obs_has_actions = True
    check_invalid = False        # humanoid-style NaN / blow-up resets

    def _setup_frames(self):
        dev, n = self.device, self.num_envs
        self.start_rot = df.quat_from_axis_angle(*self.start_axis_angle)
        self.start_rotation = tu.to_torch(self.start_rot, device=dev)
        self.inv_start_rot = tu.quat_conjugate(self.start_rotation).repeat((n, 1))
        self.x_unit_tensor = tu.to_torch([1, 0, 0], device=dev).repeat((n, 1))
        self.y_unit_tensor = tu.to_torch([0, 1, 0], device=dev).repeat((n, 1))
        self.z_unit_tensor = tu.to_torch([0, 0, 1], device=dev).repeat((n, 1))
        self.up_vec = self.y_unit_tensor.clone()
        self.heading_vec = self.x_unit_tensor.clone()
        self.basis_vec0 = self.heading_vec.clone()
        self.basis_vec1 = self.up_vec.clone()
        self.targets = tu.to_torch(list(self.target), device=dev).repeat((n, 1))
        self.start_pos = tu.to_torch([[0.0, self.start_height, 0.0]] * n, device=dev)

    def _place_root(self, builder):
        builder.joint_q[0:3] = [0.0, self.start_height, 0.0]
        builder.joint_q[3:7] = [float(v) for v in self.start_rot]

    def reset_state(self, env_ids):
        q, qd = self._q(), self._qd()
        k = len(env_ids)
        q[env_ids, 0:3] = self.start_pos[env_ids, :].clone()
        q[env_ids, 3:7] = self.start_rotation.clone()
        q[env_ids, 7:] = self.start_joint_q.clone()
        qd[env_ids, :] = 0.0
        if self.stochastic_init:
            dev = self.device
            q[env_ids, 0:3] = q[env_ids, 0:3] + 0.1 * (torch.rand(size=(k, 3), device=dev) - 0.5) * 2.0
            angle = (torch.rand(k, device=dev) - 0.5) * np.pi / 12.0
            axis = torch.nn.functional.normalize(torch.rand((k, 3), device=dev) - 0.5)
            q[env_ids, 3:7] = tu.quat_mul(q[env_ids, 3:7], tu.quat_from_angle_axis(angle, axis))
            if self.randomize_joints:
                q[env_ids, 7:] = q[env_ids, 7:] + 0.2 * (torch.rand(size=(k, self.num_joint_q - 7), device=dev) - 0.5) * 2.0
            qd[env_ids, :] = 0.5 * (torch.rand(size=(k, self.num_joint_qd), device=dev) - 0.5)
        self.actions = self.actions.clone()
        self.actions[env_ids, :] = 0.0

    def reset_noise(self):
        nq = np.zeros(self.num_joint_q, np.float32)
        nq[0:3] = 0.1 * 2.0
        if self.randomize_joints:
            nq[7:] = 0.2 * 2.0
        return nq, np.full(self.num_joint_qd, 0.5, np.float32), np.pi / 12.0

    def calculateObservations(self):
        q, qd = self._q(), self._qd()
        torso_pos, torso_rot = q[:, 0:3], q[:, 3:7]
        ang_vel = qd[:, 0:3]
        # twist (about the world origin) -> linear velocity of the torso origin
        lin_vel = qd[:, 3:6] - torch.cross(torso_pos, ang_vel, dim=-1)
        to_target = self.targets + self.start_pos - torso_pos
        to_target[:, 1] = 0.0
        target_dirs = tu.normalize(to_target)
        torso_quat = tu.quat_mul(torso_rot, self.inv_start_rot)
        up_vec = tu.quat_rotate(torso_quat, self.basis_vec1)
        heading_vec = tu.quat_rotate(torso_quat, self.basis_vec0)
        parts = [torso_pos[:, 1:2], torso_rot, lin_vel, ang_vel, q[:, 7:], self.joint_vel_obs_scaling * qd[:, 6:],
                 up_vec[:, 1:2], (heading_vec * target_dirs).sum(dim=-1).unsqueeze(-1)]
        if self.obs_has_actions:
            parts.append(self.actions.clone())
        self.obs_buf = torch.cat(parts, dim=-1)

    height_terminate = True

    def _may_reset(self):
        return (self.height_terminate or self.check_invalid or
                getattr(self, "_progress_hi", 0) > self.episode_length - 1)

    def flag_resets(self):
        self._flag_resets(self.height_terminate)

    def _locomotion_spec(self, rew_kind, act_scale, act_offset=6, act_muscle=False, **kw):
        from .. import capi
        self._act_scale_dev = act_scale.to(self.device).float().contiguous()
        isr = self.inv_start_rot[0].tolist()
        tgt = (self.targets[0] + self.start_pos[0]).tolist()
        return capi.make_env_spec(capi.ENV_LOCOMOTION, rew_kind, self.num_actions, self.num_observations,
                                  self._act_scale_dev.data_ptr(), act_offset=act_offset, act_muscle=act_muscle,
                                  obs_actions=self.obs_has_actions, inv_start_rot=isr, target_xz=(tgt[0], tgt[2]),
                                  termination_height=self.termination_height,
                                  joint_vel_obs_scaling=self.joint_vel_obs_scaling, **kw)

    def _flag_resets(self, height_terminate=True):
        if height_terminate:
            self.reset_buf = torch.where(self.obs_buf[:, 0] < self.termination_height, torch.ones_like(self.reset_buf),
                                         self.reset_buf)
        self.reset_buf = torch.where(self.progress_buf > self.episode_length - 1, torch.ones_like(self.reset_buf),
                                     self.reset_buf)
        if self.check_invalid:
            q, qd = self._q(), self._qd()
            bad = (~torch.isfinite(self.obs_buf)).any(-1) | (~torch.isfinite(q)).any(-1) | (~torch.isfinite(qd)).any(-1)
            bad = bad | (torch.abs(q) > 1e6).any(-1) | (torch.abs(qd) > 1e6).any(-1)
            self.reset_buf = torch.where(bad, torch.ones_like(self.reset_buf), self.reset_buf)
            self.rew_buf[bad] = 0.0

    def _shaped_height_reward(self):
        d = self.obs_buf[:, 0] - (self.termination_height + self.termination_tolerance)
        r = torch.clip(d, -1.0, self.termination_tolerance)
        r = torch.where(r < 0.0, -200.0 * r * r, r)
        return torch.where(r > 0.0, self.height_rew_scale * r, r)
