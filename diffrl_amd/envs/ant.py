"""Ant (reference: envs/ant.py).  9 links, 15 coordinates, 14 dofs, 25 ground contacts, 16 substeps."""
import os

import torch

from .. import dflex as df
from ..utils import load_utils as lu
from ..utils import torch_utils as tu
from .dflex_env import ASSET_DIR, find_asset
from .locomotion import FloatingBaseEnv


class AntEnv(FloatingBaseEnv):
    sim_substeps = 16
    start_height = 0.75
    termination_height = 0.27
    action_strength = 200.0
    action_penalty = 0.0
    rest_pose = [0.0, 1.0, 0.0, -1.0, 0.0, -1.0, 0.0, 1.0]

    def __init__(self, render=False, device="cuda:0", num_envs=4096, seed=0, episode_length=1000, no_grad=True,
                 stochastic_init=False, MM_caching_frequency=1, early_termination=True):
        super().__init__(num_envs, 37, 8, episode_length, MM_caching_frequency, seed, no_grad, render, device)
        self.stochastic_init = stochastic_init
        self.early_termination = early_termination
        self.height_terminate = bool(early_termination)
        self._setup_frames()
        builder = self.make_builder()
        self._place_root(builder)
        builder.joint_q[7:15] = list(self.rest_pose)
        builder.joint_target[7:15] = list(self.rest_pose)
        self.start_joint_q = tu.to_torch(self.rest_pose, device=self.device)
        self.start_joint_target = self.start_joint_q.clone()
        self._finalize(builder, ground=True)

    @staticmethod
    def make_builder():
        xml = find_asset("ant.xml")
        if xml is None:
            return df.sim.ModelBuilder.load(os.path.join(ASSET_DIR, "ant.npz"))
        b = df.sim.ModelBuilder()
        lu.parse_mjcf(xml, b, density=1000.0, stiffness=0.0, damping=1.0, contact_ke=4.e+4, contact_kd=1.e+4,
                      contact_kf=3.e+3, contact_mu=0.75, limit_ke=1.e+3, limit_kd=1.e+1, armature=0.05)
        return b

    def fused_spec(self):
        from .. import capi
        return self._locomotion_spec(capi.REW_ANT, torch.full((8,), self.action_strength),
                                     action_penalty=self.action_penalty)

    def apply_actions(self, actions):
        self.actions = actions.clone()
        self.state.joint_act.view(self.num_envs, -1)[:, 6:] = actions * self.action_strength

    def calculateReward(self):
        o = self.obs_buf
        self.rew_buf = (o[:, 5] + 0.1 * o[:, 27] + o[:, 28] + (o[:, 0] - self.termination_height)
                        + torch.sum(self.actions ** 2, dim=-1) * self.action_penalty)
        self._flag_resets(height_terminate=self.early_termination)
