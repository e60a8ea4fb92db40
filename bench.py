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


def measure_other_config(name, n, H, mm, device, steps=2):
    """fwd+adjoint env-steps/s of another BASELINE.json configuration on this GPU: the same rollout (H x DFlexEnv.step, loss =
    -sum(rew), one backward), captured as a HIP graph, `steps` timed replays after one warm-up; adjoint / forward kernel
    times by HIP events.  Informational: the headline `value` is the workload named in config.workload."""
    saved = MM_FREQ[name]
    MM_FREQ[name] = mm
    try:
        from diffrl_amd.graph import GraphedRollout
        env = make_env(name, n, str(device))
        gen = torch.Generator().manual_seed(1)
        actions = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0).to(device)
        acts = actions.detach().clone().requires_grad_(True)

        def body(e):
            e.initialize_trajectory()
            return reward_loss([e.step(a_t)[1] for a_t in acts.unbind(0)])

        env.clear_grad()
        env.reset()
        roll = GraphedRollout(env, body, leaves=[acts], carry_state=False)
        roll.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            roll.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert torch.isfinite(acts.grad).all()
        t_bwd = time_backward_kernel(env, name, n, H, 10, device)
        eng = env.model.engine()
        return {"workload": "%s %d envs x H=%d, MM_caching_frequency %d" % (name, n, H, mm), "value": steps * n * H / el,
                "unit": "env-steps/s", "ms_per_rollout": el / steps * 1e3, "kernel_ms": t_bwd * 1e3,
                "fwd_kernel_ms": time_backward_kernel.fwd_s * 1e3,
                "ckpt_bytes_per_env_step": 4 * int(eng._lib.dsim_ckpt_floats_mm(eng._h, env.sim_substeps, mm))}
    except Exception as ex:
        return {"workload": "%s %d envs x H=%d" % (name, n, H), "value": None, "error": str(ex)[:200]}
    finally:
        MM_FREQ[name] = saved


def cpu_baseline(name, budget_s=12.0):
    """CPU baseline beside the GPU number (oracle/cpu_baseline.py, a subprocess so that nothing GPU-related is forked):
    the reference's own CPU path when its checkout is present (kind "reference"), otherwise the multi-threaded scalar
    port oracle/dsim_oracle.cpp (kind "port") with the recorded reference figure attached."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), name, str(budget_s)],
                         capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        return {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: " + out.stderr[-200:]}
    return json.loads(line[-1])


def csrc_hash():
    """identifies the kernel sources a measured figure belongs to (roofline.traffic is read from a committed PMC file)"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "diffrl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--env", default="ant")
    ap.add_argument("--envs-per-gpu", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--mm-freq", type=int, default=0, help="MM_caching_frequency (0: the examples/cfg/shac value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short measurements of the other BASELINE.json configurations (Humanoid 1024 x 32, SNUHumanoid "
                         "512 x 32, Ant with MM_caching_frequency 1) that the default single-GPU Ant run appends as `other_configs`")
    ap.add_argument("--eager", action="store_true", help="time the Python-driven step loop instead of the graph replay")
    ap.add_argument("--strict", action="store_true", help="exit non-zero if the graph capture fell back to the eager loop")
    ap.add_argument("--launcher", action="store_true",
                    help="go through torch.distributed.run even for --gpus 1 (an RCCL group of one rank)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch the ranks, rendezvous, exchange the timing all-reduce and print the JSON skeleton, "
                         "without touching the GPU engine (CPU test of the launcher; backend gloo when no GPU is visible)")
    return ap.parse_args(argv)


def launch_ranks(a, argv):
    """`python bench.py --gpus N` run directly (no torch.distributed.run around it): become the launcher.
    One process per GPU, rendezvous on 127.0.0.1; returns the launcher's exit code."""
    import socket
    import subprocess
    if not a.dry_run or torch.cuda.is_available():
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible on this node; refusing to "
                             "report a smaller job under the requested label\n" % (a.gpus, have))
            return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    if a.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and (a.gpus > 1 or a.launcher or a.dry_run):
        return launch_ranks(a, argv)

    from diffrl_amd import sharding
    rank, local, world = sharding.world()
    if world != a.gpus:
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d; the label must match the job\n" % (world, a.gpus))
        return 2
    dist = under_launcher   # launched by torch.distributed.run: RCCL process group even for one rank
    use_gpu = torch.cuda.is_available()
    if not use_gpu and not a.dry_run:
        sys.stderr.write("bench.py: no GPU visible; this benchmark has no CPU path\n")
        return 2
    if use_gpu:
        if torch.cuda.device_count() <= local:
            sys.stderr.write("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)\n" % (rank, local, torch.cuda.device_count()))
            return 2
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
    else:
        device = torch.device("cpu")
    rccl_ranks = 1
    if dist:
        import torch.distributed as td
        sharding.init("nccl" if use_gpu else "gloo", device if use_gpu else None)
        rccl_ranks = td.get_world_size()
        assert rccl_ranks == a.gpus

    def barrier():
        if dist:
            td.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    n, H = a.envs_per_gpu, a.horizon
    mm = a.mm_freq or MM_FREQ[a.env]
    lo, hi = sharding.shard_range(n * world, rank, world)   # this rank's env-index range of the whole job
    if a.dry_run:
        barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))
        barrier()
        el = sharding.max_over_ranks(time.perf_counter() - t0, device)
        owned = sharding.sum_over_ranks(hi - lo, device)
        if rank == 0:
            print(json.dumps({"metric": "fwd+adjoint env-steps/sec", "dry_run": True, "value": None, "unit": "env-steps/s",
                              "n_gpus": world, "rccl_ranks": rccl_ranks, "backend": "nccl" if use_gpu else "gloo",
                              "steps": a.steps, "warmup": a.warmup, "scaling": "weak", "envs_total": int(owned),
                              "slowest_rank_s": el,
                              "config": {"workload": "%s %d envs/GPU x H=%d" % (a.env, n, H), "envs_per_gpu": n}}))
        if dist:
            td.barrier()
            td.destroy_process_group()
        return 0

    MM_FREQ[a.env] = mm
    env = make_env(a.env, n, str(device))
    gen = torch.Generator().manual_seed(1 + rank)
    actions = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0).to(device)

    import gc
    submission = "eager: one launch per env.step each way, issued from the Python loop"
    fallback = False
    roll = None
    if not a.eager:
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
            submission = "one HIP graph per rollout: the %d forward + %d adjoint launches captured through DFlexEnv.step" % (H, H)
        except Exception as ex:  # capture unsupported on this stack: measure the eager loop instead, and say so LOUDLY
            roll = None
            fallback = True
            submission = "eager (graph capture failed: %s)" % str(ex)[:120]
            sys.stderr.write("bench.py: WARNING graph capture failed, timing the eager loop instead: %s\n" % str(ex)[:300])

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
        # HBM bytes of one adjoint launch.  Measured: the committed PMC passes (rocprofv3 cannot run inside this process),
        # valid only for the kernel sources they were taken at (csrc hash) and for that env / N.  Otherwise the analytic
        # figure: the adjoint reads its whole checkpoint (dsim_ckpt_floats_mm floats per environment) plus the boundary tensors.
        eng = env.model.engine()
        ckpt_floats = int(eng._lib.dsim_ckpt_floats_mm(eng._h, env.sim_substeps, mm))
        traffic_analytic = 4 * n * ckpt_floats + bwd_bytes
        traffic, traffic_src = traffic_analytic, "analytic: checkpoint words x N x 4 + boundary tensors"
        try:
            pmcs = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json"))
            for f in reversed(pmcs):
                pmc = json.load(open(os.path.join(ROOT, "profiles", f)))
                if (pmc.get("env", "ant") == a.env and pmc.get("n_envs", 1024) == n and pmc.get("mm_freq", MM_FREQ[a.env]) == mm
                        and pmc.get("csrc_hash") == csrc_hash()):
                    traffic, traffic_src = pmc["traffic_bytes_per_launch"], "rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), profiles/" + f
                    break
        except Exception:
            pass
        out = {
            "metric": "fwd+adjoint env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %d envs/GPU x H=%d through DFlexEnv.step, loss=-sum(rew), 1 backward"
                                   % (a.env, n, H), "envs_per_gpu": n, "envs_total": n * world, "horizon": H,
                       "substeps": env.sim_substeps,
                       "mm_freq": mm, "sharding": "envs by index, no collective", "submission": submission,
                       "submission_fallback": fallback},
            "eager_env_steps_per_s": eager_value,
            "roofline": {"bound": "hbm", "kernel": "dsim_env_bwd_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "csrc_hash": csrc_hash(), "ckpt_bytes_per_env_step": 4 * ckpt_floats,
                         "ckpt_bytes_per_rollout": 4 * ckpt_floats * n * H,
                         "kernel_ms": t_bwd * 1e3, "fwd_kernel_ms": time_backward_kernel.fwd_s * 1e3,
                         "alg_bytes_per_launch": bwd_bytes,
                         "note": "fused kernel is VALU/latency-bound by construction (SURVEY 8d); traffic >> algorithmic bytes on purpose: "
                                 "the saved forward block the adjoint reads back instead of recomputing, see DESIGN.md section 4"},
            # fp32 vector-ALU view of the same launch pair (SURVEY 8d asks for it next to the HBM fraction): ~1.2 MFLOP per
            # Ant env-step fwd+adjoint (SURVEY's op-count estimate) against the 157.3 TFLOP/s fp32 vector peak
            "fp32_valu_frac_est": (1.2e6 * n / (t_bwd + time_backward_kernel.fwd_s)) / 157.3e12 if a.env == "ant" else None,
        }
        # forward-only serving path (dflex.config.no_grad: no checkpoint traffic), SURVEY.md 8(f).4 -- informational
        with torch.no_grad():
            spec = env._spec()
            q, qd = env.state.joint_q.detach().clone(), env.state.joint_qd.detach().clone()
            for _ in range(5):
                eng.env_forward(spec, q, qd, actions[0], env.sim_dt, env.sim_substeps, mm, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(100):
                q, qd, _, _, _ = eng.env_forward(spec, q, qd, actions[t % H], env.sim_dt, env.sim_substeps, mm, False)
            torch.cuda.synchronize()
            out["no_grad_forward_env_steps_per_s"] = 100 * n / (time.perf_counter() - t0)
        if not a.no_other_configs and world == 1 and a.env == "ant" and not a.eager:
            # BASELINE.json configs[2], configs[3] and the MM_caching_frequency = 1 variant of configs[1] (SURVEY.md 8(d):
            # "also report 1"), each a few seconds: driver-visible numbers next to the headline
            del env, roll
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = [measure_other_config("humanoid", 1024, H, MM_FREQ["humanoid"], device),
                                    measure_other_config("snu", 512, H, MM_FREQ["snu"], device),
                                    measure_other_config("ant", n, H, 1, device)]
        if not a.no_cpu_baseline and world == 1:   # reported at N = 1 only (the host cores are the same for every N)
            out["cpu_baseline"] = cpu_baseline(a.env)
        print(json.dumps(out))
    if dist:
        td.barrier()   # rank 0 is still measuring its extras (eager loop, CPU baseline): tear the group down together
        td.destroy_process_group()
    if fallback and a.strict:
        sys.stderr.write("bench.py: --strict and the graph capture fell back to the eager loop\n")
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
