"""Batched quaternion helpers for the observation / reward code (quaternions are (x, y, z, w)).
Same function names as the reference's utils/torch_utils.py so environment code reads the same."""
import numpy as np
import torch


def to_torch(x, dtype=torch.float, device="cuda:0", requires_grad=False):
    return torch.tensor(np.asarray(x), dtype=dtype, device=device, requires_grad=requires_grad)


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_conjugate(q):
    return torch.cat((-q[..., :3], q[..., 3:]), dim=-1)


def quat_mul(a, b):
    """Hamilton product a (x) b."""
    av, aw = a[..., :3], a[..., 3:]
    bv, bw = b[..., :3], b[..., 3:]
    v = aw * bv + bw * av + torch.cross(av, bv, dim=-1)
    w = aw * bw - (av * bv).sum(-1, keepdim=True)
    return torch.cat((v, w), dim=-1)


def quat_rotate(q, v):
    qv, qw = q[..., :3], q[..., 3:]
    return v * (2.0 * qw * qw - 1.0) + torch.cross(qv, v, dim=-1) * qw * 2.0 + qv * (qv * v).sum(-1, keepdim=True) * 2.0


def quat_from_angle_axis(angle, axis):
    half = (angle * 0.5).unsqueeze(-1)
    return normalize(torch.cat((normalize(axis) * half.sin(), half.cos()), dim=-1))


def normalize_angle(x):
    return torch.atan2(torch.sin(x), torch.cos(x))
