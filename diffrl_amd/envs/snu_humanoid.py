"""SNUHumanoid, lower body with 152 muscle-tendon units (reference: envs/snu_humanoid.py).
11 links (free root, ball + revolute joints), 29 coordinates, 24 dofs, 88 box-corner contacts."""
import math
import os

import torch

from .. import dflex as df
from ..utils import load_utils as lu
from ..utils import torch_utils as tu
from .dflex_env import ASSET_DIR, find_asset
from .locomotion import FloatingBaseEnv


class SNUHumanoidEnv(FloatingBaseEnv):
    sim_substeps = 48
    start_height = 1.0
    start_axis_angle = ((0.0, 1.0, 0.0), math.pi * 0.5)
    termination_height = 0.46
    termination_tolerance = 0.05
    height_rew_scale = 4.0
    action_penalty = -0.001
    str_scale = 0.6
    randomize_joints = False
    obs_has_actions = False
    sanitize_grads = True
    check_invalid = True
    segments = {"Pelvis", "FemurR", "TibiaR", "TalusR", "FootThumbR", "FootPinkyR", "FemurL", "TibiaL", "TalusL",
                "FootThumbL", "FootPinkyL"}

    def __init__(self, render=False, device="cuda:0", num_envs=4096, seed=0, episode_length=1000, no_grad=True,
                 stochastic_init=False, MM_caching_frequency=1):
        self.num_muscles = 152
        super().__init__(num_envs, 53, self.num_muscles, episode_length, MM_caching_frequency, seed, no_grad, render,
                         device)
        self.stochastic_init = stochastic_init
        self._setup_frames()
        builder, f0 = self.make_builder()
        builder.joint_q[1] = self.start_height
        builder.joint_q[3:7] = [float(v) for v in self.start_rot]
        self.start_pos = tu.to_torch([[builder.joint_q[0], self.start_height, builder.joint_q[2]]] * num_envs,
                                     device=self.device)
        self.start_joint_q = tu.to_torch(builder.joint_q[7:], device=self.device)
        self.start_joint_target = self.start_joint_q.clone()
        # the reference scales f0 by str_scale twice (snu_humanoid.py:173-177)
        self.muscle_strengths = tu.to_torch([self.str_scale * self.str_scale * f for f in f0],
                                            device=self.device).repeat(num_envs)
        self._finalize(builder, ground=True)

    @classmethod
    def make_builder(cls):
        skel, musc = find_asset("snu/human.xml"), find_asset("snu/muscle284.xml")
        if skel is None or musc is None:
            b = df.sim.ModelBuilder.load(os.path.join(ASSET_DIR, "snu_humanoid.npz"))
            return b, [p[0] for p in b.muscle_params]
        b = df.sim.ModelBuilder()
        s = lu.Skeleton(skel, musc, b, cls.segments, stiffness=5.0, damping=2.0, contact_ke=5e3, contact_kd=2e3,
                        contact_kf=1e3, contact_mu=0.5, limit_ke=1e3, limit_kd=1e1, armature=0.05)
        return b, [m.muscle_strength for m in s.muscles]

    def fused_spec(self):
        from .. import capi
        return self._locomotion_spec(capi.REW_SNU, self.muscle_strengths[:self.num_muscles], act_offset=0,
                                     act_muscle=True, action_penalty=self.action_penalty)

    def stored_actions(self, actions):
        return torch.clip(actions, -1.0, 1.0) * 0.5 + 0.5

    def apply_actions(self, actions):
        actions = actions * 0.5 + 0.5
        self.actions = actions.clone()
        self.model.muscle_activation = actions.view(-1) * self.muscle_strengths

    def calculateReward(self):
        o = self.obs_buf
        act_penalty = torch.sum(torch.abs(self.actions), dim=-1) * self.action_penalty
        self.rew_buf = o[:, 5] + 0.1 * o[:, 51] + o[:, 52] + act_penalty
        self._flag_resets()
