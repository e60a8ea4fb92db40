"""GPU: launch-time guard of the three BASELINE configurations.  The kernels' speed rests on instruction scheduling that only A/B
runs used to watch (CHANGELOG "measured and rejected"): a toolchain or source change that costs a launch
15 % more -- a phase state landing in scratch, a lost helper wavefront, a register row spilled -- fails HERE instead of
surfacing in the next bench.  Bounds: 1.15 x the recorded launch times (box-to-box spread over rounds 5
and 6 was 1 %), measured as bench.py measures them (HIP events around the env-step launches, same shapes as the rollout), plus the
headline itself: the Ant 1024 x 32 rollout graph replayed ten times must hold HEADLINE_FLOOR env-steps/s."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# env, environments, (forward, adjoint) launch in ms at round 6 (profiles/r06_bench_final.json; round 5: 0.0524 / 0.0529, 0.1818 / 0.1801,
# 0.2230 / 0.2339)
RECORDED = [("ant", 1024, 0.0524, 0.0530), ("humanoid", 1024, 0.1817, 0.1762), ("snu", 512, 0.2211, 0.2329)]
MARGIN = 1.15
HEADLINE_FLOOR = 9.2e6   # env-steps/s, Ant 1024 envs x H=32, forward + adjoint through DFlexEnv.step (bench.py's timed submission)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n,fwd_ms,bwd_ms", RECORDED)
def test_env_step_launches_stay_within_the_recorded_times(name, n, fwd_ms, bwd_ms):
    if os.environ.get("DSIM_LIB") or os.environ.get("DSIM_FORCE_GENERIC"):
        pytest.skip("developer override in the environment: the bounds belong to the shipped specialised kernels")
    import bench
    device = torch.device("cuda:0")
    env = bench.make_env(name, n, str(device))
    assert env.model.engine().variant > 0, "the specialised kernel set of %s was not selected" % name
    best = [1e9, 1e9]
    for _ in range(3):   # minimum of three: a neighbour on the box must not fail the guard
        rf = bench.roofline_record(env, name, n, 32, bench.MM_FREQ[name], device, 20, counters=False)
        best = [min(best[0], rf["fwd_kernel_ms"]), min(best[1], rf["bwd_kernel_ms"])]
    print("%s %d: forward %.4f ms (recorded %.4f), adjoint %.4f ms (recorded %.4f)" % (name, n, best[0], fwd_ms, best[1], bwd_ms))
    assert best[0] < MARGIN * fwd_ms, "forward launch %.4f ms, recorded %.4f" % (best[0], fwd_ms)
    assert best[1] < MARGIN * bwd_ms, "adjoint launch %.4f ms, recorded %.4f" % (best[1], bwd_ms)


@pytest.mark.gpu
def test_headline_graph_replays_hold_the_floor():
    """the timed submission of bench.py (one HIP graph per rollout: 32 x DFlexEnv.step + the backward sweep), ten replays, best of
    three timings: a toolchain or source regression of the headline fails the GPU tier instead of surfacing in BENCH"""
    if os.environ.get("DSIM_LIB") or os.environ.get("DSIM_FORCE_GENERIC"):
        pytest.skip("developer override in the environment: the floor belongs to the shipped specialised kernels")
    import time

    import bench
    from diffrl_amd.graph import GraphedRollout
    device = torch.device("cuda:0")
    n, H = 1024, 32
    env = bench.make_env("ant", n, str(device))
    gen = torch.Generator().manual_seed(1)
    acts = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0).to(device).requires_grad_(True)

    def body(e):
        e.initialize_trajectory()
        return bench.reward_loss([e.step(a_t)[1] for a_t in acts.unbind(0)])

    env.clear_grad()
    env.reset()
    roll = GraphedRollout(env, body, leaves=[acts], carry_state=False)
    for _ in range(3):
        roll.replay()
    best = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            roll.replay()
        torch.cuda.synchronize()
        best = max(best, 10 * n * H / (time.perf_counter() - t0))
    assert torch.isfinite(acts.grad).all()
    print("ant 1024 x 32 graph replays: %.3f M env-steps/s (floor %.1f M)" % (best / 1e6, HEADLINE_FLOOR / 1e6))
    assert best >= HEADLINE_FLOOR, "headline %.3f M env-steps/s below the floor %.1f M" % (best / 1e6, HEADLINE_FLOOR / 1e6)
