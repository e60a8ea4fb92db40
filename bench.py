"""Headline benchmark: forward + adjoint env-steps/s, Ant 1024 envs x H=32 per GPU (BASELINE.json).

One bench "step" = one rollout through the DFlexEnv surface: H env.step() calls with fixed synthetic
actions, loss = -sum(reward), one backward through all H steps (what algorithms/shac.py:169-300 +
shac.py:411 do, minus the actor network).  N GPUs = N processes (torch.distributed.run), each with its own
shard of environments; no collective touches the data path (the all_reduce below only combines timings).

The timed submission is a HIP-graph replay of that rollout (captured once through DFlexEnv.step with
diffrl_amd/graph.py, checked bit for bit against the eager loop; every replay re-executes all launches); the same
rollouts driven step by step from Python are reported as `eager_env_steps_per_s` (`--eager` times those instead).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the adjoint kernel), `cpu_baseline`
times the scalar CPU oracle on a bounded sample of the same workload on the box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ALG_BYTES = {"ant": 748, "humanoid": 1424, "snu": 2884, "cartpole": 104, "hopper": 300, "cheetah": 444}  # per env-step fwd+adjoint, SURVEY.md 8(d)
HBM_PEAK_GBS = 8000.0
MM_FREQ = {"ant": 16, "humanoid": 48, "snu": 8, "cartpole": 4, "hopper": 16, "cheetah": 16}  # examples/cfg/shac/*.yaml


def make_env(name, n, device):
    from diffrl_amd import envs
    cls = {"ant": envs.AntEnv, "humanoid": envs.HumanoidEnv, "snu": envs.SNUHumanoidEnv,
           "cartpole": envs.CartPoleSwingUpEnv, "hopper": envs.HopperEnv, "cheetah": envs.CheetahEnv}[name]
    kw = dict(num_envs=n, device=device, render=False, seed=0, episode_length=100000, no_grad=False,
              stochastic_init=False, MM_caching_frequency=MM_FREQ[name])
    if name in ("ant", "cartpole", "hopper", "cheetah"):
        kw["early_termination"] = False
    return cls(**kw)


def rollout(env, actions):
    """one SHAC-style trajectory: H x env.step on fresh environments, loss = -sum of rewards, one backward"""
    env.clear_grad()
    env.reset()
    env.initialize_trajectory()
    acts = actions.detach().requires_grad_(True)
    rews = []
    for a_t in acts.unbind(0):
        obs, rew, done, info = env.step(a_t)
        rews.append(rew)
    loss = reward_loss(rews)
    loss.backward()
    return acts.grad


_W = {}


def reward_loss(rews):
    """-sum_t sum_env w[t, env] * rew[t, env] with w = 1 (SHAC weights the steps by gamma^t, algorithms/shac.py:215-216).
    Written with an explicit weight tensor so that every step receives a contiguous reward cotangent (a bare .sum()
    hands autograd a stride-0 broadcast that each of the H backward steps would first have to materialise)."""
    r = torch.stack(rews)
    key = (r.shape, r.device)
    if key not in _W:
        _W[key] = torch.ones_like(r)
    return -(r * _W[key]).sum()


def time_backward_kernel(env, name, n, H, reps, device):
    """average duration of the adjoint kernel, HIP events on the launch stream, same shapes as the rollout"""
    eng = env.model.engine()
    spec = env._spec()
    S, mm = env.sim_substeps, MM_FREQ[name]
    q = env.state.joint_q.detach().clone()
    qd = env.state.joint_qd.detach().clone()
    acts = torch.zeros((n, env.num_actions), device=device)
    qo, qdo, obs, rew, ck = eng.env_forward(spec, q, qd, acts, env.sim_dt, S, mm, True)
    gq, gqd, go, gr = torch.randn_like(q), torch.randn_like(qd), torch.randn_like(obs), torch.randn_like(rew)
    for _ in range(3):
        eng.env_backward(spec, ck, acts, env.sim_dt, S, mm, gq, gqd, go, gr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        eng.env_backward(spec, ck, acts, env.sim_dt, S, mm, gq, gqd, go, gr)
    e1.record()
    torch.cuda.synchronize()
    t_bwd = e0.elapsed_time(e1) / reps * 1e-3
    # the forward launch with checkpoint, same shapes (reported next to the adjoint's figure)
    for _ in range(3):
        eng.env_forward(spec, q, qd, acts, env.sim_dt, S, mm, True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        eng.env_forward(spec, q, qd, acts, env.sim_dt, S, mm, True)
    e1.record()
    torch.cuda.synchronize()
    time_backward_kernel.fwd_s = e0.elapsed_time(e1) / reps * 1e-3
    return t_bwd


def cpu_baseline(name, budget_s=12.0):
    """scalar CPU oracle (oracle/dsim_oracle.cpp, a port of the reference's CPU path) on a bounded sample, one process
    per host core (up to 32); run as a subprocess so that nothing GPU-related is forked"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), name, str(budget_s)],
                         capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        return {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: " + out.stderr[-200:]}
    return json.loads(line[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--env", default="ant")
    ap.add_argument("--envs-per-gpu", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="time the Python-driven step loop instead of the graph replay")
    a = ap.parse_args()

    from diffrl_amd import sharding
    rank, local, world = sharding.world()
    dist = world > 1 or "RANK" in os.environ   # launched by torch.distributed.run: RCCL process group even for one rank
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if dist:
        import torch.distributed as td
        sharding.init("nccl", device)

    n, H = a.envs_per_gpu, a.horizon
    env = make_env(a.env, n, str(device))
    gen = torch.Generator().manual_seed(1 + rank)
    actions = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0).to(device)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    import gc
    ap_eager = a.eager
    submission = "eager: one launch per env.step each way, issued from the Python loop"
    roll = None
    if not ap_eager:
        # whole rollout (32 x DFlexEnv.step + the backward sweep) captured once as a HIP graph, replayed as one submission
        # (diffrl_amd/graph.py).  Every replay re-executes all 64 launches on the same inputs.
        try:
            from diffrl_amd.graph import GraphedRollout
            g_eager = rollout(env, actions).clone()
            acts = actions.detach().clone().requires_grad_(True)

            def body(e):
                e.initialize_trajectory()
                rews = [e.step(a_t)[1] for a_t in acts.unbind(0)]
                return reward_loss(rews)

            env.clear_grad()
            env.reset()
            roll = GraphedRollout(env, body, leaves=[acts], carry_state=False)
            roll.replay()
            torch.cuda.synchronize()
            assert torch.equal(acts.grad, g_eager), "graph replay and eager rollout disagree"
            submission = "one HIP graph per rollout: the 32 forward + 32 adjoint launches captured through DFlexEnv.step"
        except Exception as ex:  # capture unsupported on this stack: measure the eager loop instead, and say so
            roll = None
            submission = "eager (graph capture failed: %s)" % str(ex)[:120]

    def one():
        if roll is not None:
            roll.replay()
            return acts.grad
        return rollout(env, actions)

    for _ in range(a.warmup):
        one()
    gc.collect()
    gc.disable()   # no cyclic-GC pause inside the timed region (a gen-2 collection costs tens of ms once every few rollouts)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        grad = one()
    barrier()
    el = time.perf_counter() - t0
    gc.enable()
    assert torch.isfinite(grad).all()
    el = sharding.max_over_ranks(el, device)
    total_env_steps = a.steps * world * n * H
    value = total_env_steps / el
    eager_value = None
    if roll is not None and rank == 0:
        # the same rollouts driven step by step from Python (what a caller that does not capture gets)
        env2 = make_env(a.env, n, str(device))
        for _ in range(2):
            rollout(env2, actions)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            rollout(env2, actions)
        torch.cuda.synchronize()
        eager_value = 5 * n * H / (time.perf_counter() - t0)
        del env2

    if rank == 0:
        t_bwd = time_backward_kernel(env, a.env, n, H, 50, device)
        # algorithmic bytes of ONE adjoint launch: re-read (q,qd,act) + read (gq',gqd') + write (gq,gqd,gact)
        # (the fused kernel additionally reads the obs/reward cotangents; not counted, SURVEY's figure is kept)
        nq, nd = env.num_joint_q, env.num_joint_qd
        na_in = env.model.muscles_per_articulation if env.model.muscle_count else nd
        bwd_bytes = 4 * n * ((nq + nd + na_in) + (nq + nd) + (nq + nd + na_in))
        achieved = bwd_bytes / t_bwd / 1e9
        traffic = None  # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        try:
            pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_final_pmc.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))
            if a.env == "ant" and n == 1024 and pmc.get("kernel") in ("dsim_bwd_kernel", "dsim_env_bwd_kernel"):
                traffic = pmc["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "fwd+adjoint env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %d envs/GPU x H=%d through DFlexEnv.step, loss=-sum(rew), 1 backward"
                                   % (a.env, n, H), "envs_per_gpu": n, "horizon": H, "substeps": env.sim_substeps,
                       "mm_freq": MM_FREQ[a.env], "sharding": "envs by index, no collective", "submission": submission},
            "eager_env_steps_per_s": eager_value,
            "roofline": {"bound": "hbm", "kernel": "dsim_env_bwd_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": t_bwd * 1e3, "fwd_kernel_ms": time_backward_kernel.fwd_s * 1e3,
                         "alg_bytes_per_launch": bwd_bytes,
                         "note": "fused kernel is VALU/latency-bound by construction (SURVEY 8d); traffic >> algorithmic bytes on purpose: "
                                 "39.3 MB of it is the saved forward block the adjoint reads back instead of recomputing "
                                 "(measured 19 % faster; HBM is at ~3 % of peak either way), see DESIGN.md section 4"},
            # fp32 vector-ALU view of the same launch pair (SURVEY 8d asks for it next to the HBM fraction): ~1.2 MFLOP per
            # Ant env-step fwd+adjoint (SURVEY's op-count estimate) against the 157.3 TFLOP/s fp32 vector peak
            "fp32_valu_frac_est": (1.2e6 * n / (t_bwd + time_backward_kernel.fwd_s)) / 157.3e12 if a.env == "ant" else None,
        }
        # forward-only serving path (dflex.config.no_grad: no checkpoint traffic), SURVEY.md 8(f).4 -- informational
        with torch.no_grad():
            eng, spec = env.model.engine(), env._spec()
            q, qd = env.state.joint_q.detach().clone(), env.state.joint_qd.detach().clone()
            for _ in range(5):
                eng.env_forward(spec, q, qd, actions[0], env.sim_dt, env.sim_substeps, MM_FREQ[a.env], False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(100):
                q, qd, _, _, _ = eng.env_forward(spec, q, qd, actions[t % H], env.sim_dt, env.sim_substeps, MM_FREQ[a.env], False)
            torch.cuda.synchronize()
            out["no_grad_forward_env_steps_per_s"] = 100 * n / (time.perf_counter() - t0)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.env)
        print(json.dumps(out))
    if dist:
        td.barrier()   # rank 0 is still measuring its extras (eager loop, CPU baseline): tear the group down together
        td.destroy_process_group()


if __name__ == "__main__":
    main()
