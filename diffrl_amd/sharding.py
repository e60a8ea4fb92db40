"""Environment sharding across the GPUs of one node.

Environments are independent (no cross-env term anywhere in the step, contacts are env-vs-ground only,
dflex/dflex/model.py:444-447), so the multi-GPU path is: one process per GPU, contiguous env-index
ranges, NO collective in the simulation step.  `torch.distributed` is only used to agree on timings /
totals (backend "nccl" == RCCL on the GPUs, "gloo" in the CPU tests)."""
import os

import torch


def world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(total_envs, rank, world_size):
    """Contiguous [lo, hi) env-index range of `rank`; sizes differ by at most one, every env owned once."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(total_envs, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend, device=None):
    import torch.distributed as td
    if td.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    td.init_process_group(backend, **kw)


def max_over_ranks(value, device="cpu"):
    """Slowest rank's elapsed time (the job's time)."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(values, device="cpu"):
    """[world][len(values)] floats: every rank's figures on every rank (straggler diagnosis of a multi-GPU job: which rank was
    slow, not only that one was).  One all_gather of a few doubles; a single process returns [values]."""
    import torch.distributed as td
    vals = [float(v) for v in values]
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return [vals]
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(td.get_world_size())]
    td.all_gather(out, t)
    return [o.cpu().tolist() for o in out]


def gather_strings(text):
    """[world] strings (device names of the ranks); host-side all_gather_object"""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return [text]
    out = [None] * td.get_world_size()
    td.all_gather_object(out, text)
    return out
