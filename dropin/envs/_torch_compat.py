"""Compatibility shim for running the UNMODIFIED reference callers on a current PyTorch.

The reference (written for torch 1.x) keeps some per-environment bookkeeping tensors on the CPU and indexes them with index
tensors that live on the GPU -- `self.episode_length = torch.zeros(self.num_envs, dtype=int)` indexed by
`done_env_ids = done.nonzero()` (algorithms/shac.py:156, 277-288; the same pattern in evaluate_policy, :308-338).  torch 1.x
moved such indices to the host implicitly; torch 2.x raises "indices should be either on cpu or on the same device as the
indexed tensor".  This module restores the old behaviour for CPU tensors only (GPU tensors take the unmodified path), so
that no reference file has to be edited.  It is installed by `import envs` through THIS drop-in package only; the
`diffrl_amd` package itself never patches torch."""
import torch


def install():
    if getattr(torch.Tensor, "_dsim_legacy_cpu_indexing", False):
        return
    _get, _set = torch.Tensor.__getitem__, torch.Tensor.__setitem__

    def _host(i):
        return i.cpu() if isinstance(i, torch.Tensor) and i.device.type != "cpu" else i

    def _fix(self, idx):
        if self.device.type != "cpu":
            return idx
        if isinstance(idx, tuple):
            return tuple(_host(i) for i in idx)
        return _host(idx)

    def __getitem__(self, idx):
        return _get(self, _fix(self, idx))

    def __setitem__(self, idx, val):
        return _set(self, _fix(self, idx), val)

    torch.Tensor.__getitem__ = __getitem__
    torch.Tensor.__setitem__ = __setitem__
    torch.Tensor._dsim_legacy_cpu_indexing = True
