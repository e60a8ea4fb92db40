"""CPU-only, world_size 2 over gloo: the N>1 launcher logic (env-index shards, timing reduction, whole-job
throughput accounting) that bench.py uses with RCCL on the GPUs.  No collective is on the data path."""
import os

import pytest
import torch
import torch.multiprocessing as mp

from diffrl_amd import sharding


def test_shard_ranges_partition():
    for total, w in [(1024, 1), (1024, 8), (8192, 8), (10, 3), (7, 8)]:
        seen = []
        for r in range(w):
            lo, hi = sharding.shard_range(total, r, w)
            assert 0 <= lo <= hi <= total
            seen += list(range(lo, hi))
        assert seen == list(range(total))
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sharding.init("gloo")
    r, lr, w = sharding.world()
    lo, hi = sharding.shard_range(1000, r, w)
    # each rank "steps" its own envs; elapsed differs per rank; job time = max, job work = sum
    elapsed = 1.0 + 0.5 * r
    job_t = sharding.max_over_ranks(elapsed)
    job_steps = sharding.sum_over_ranks((hi - lo) * 32)
    # every rank builds its shard with the same seed and local environment indices 0..n-1: the key of the in-kernel restart
    # noise must still differ between ranks (DFlexEnv._philox_key mixes the rank in), or all shards would draw the same restarts
    from diffrl_amd import envs
    e = envs.CartPoleSwingUpEnv(num_envs=4, device="cpu", no_grad=True, seed=7)
    q.put((r, lo, hi, job_t, job_steps, e._philox_key()))
    import torch.distributed as td
    td.barrier()
    td.destroy_process_group()


def test_two_process_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 500), (500, 1000)]
    assert all(abs(r[3] - 1.5) < 1e-12 for r in res)          # max over ranks
    assert all(abs(r[4] - 32000) < 1e-9 for r in res)         # whole-job env-steps
    assert res[0][5] != res[1][5] and all(0 <= r[5] < 2 ** 64 for r in res)   # rank-specific Philox keys
