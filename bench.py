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


def pmc_record(name, n, mm):
    """the committed counter record (profiles/*_pmc.json, tools/profile.sh + tools/make_profile_record.py) that was measured
    at THESE kernel sources (csrc hash), this environment, N and mass-matrix frequency -- or None.  rocprofv3 cannot run
    inside this process, so counters are quoted from the tracked files and only when they describe the kernels being timed."""
    try:
        h = csrc_hash()
        for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json")), reverse=True):
            pmc = json.load(open(os.path.join(ROOT, "profiles", f)))
            if (pmc.get("env", "ant") == name and pmc.get("n_envs", 1024) == n and pmc.get("mm_freq", MM_FREQ[name]) == mm
                    and pmc.get("csrc_hash") == h):
                pmc["file"] = "profiles/" + f
                return pmc
    except Exception:
        pass
    return None


def census_record():
    """profiles/opcode_census.json (tools/opcode_census.py: static census of the kernels' substep loops -- the share of VALU
    instructions that carries floating-point work and the flops each carries), if it was taken at THESE kernel sources"""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "opcode_census.json")))
        return c if c.get("csrc_hash") == csrc_hash() else None
    except Exception:
        return None


KERNEL_TAG = {"ant": "Ant", "humanoid": "Humanoid", "snu": "Snu", "cartpole": "Cartpole", "hopper": "Hopper", "cheetah": "Cheetah"}


def census_flops_per_inst(census, kernel, name, waves_per_env):
    """flops per VALU instruction of the kernel variant the counters describe (helper-wave kernels: 2 waves per environment,
    pair kernels: half a wave, otherwise the plain mapping)"""
    if not census or waves_per_env is None:
        return None
    mode = "helper" if abs(waves_per_env - 2.0) < 1e-6 else ("pair" if abs(waves_per_env - 0.5) < 1e-6 else "plain")
    k = census["kernels"].get("%s<%s,%s>" % (kernel, KERNEL_TAG.get(name, "generic"), mode))
    return k["flops_per_valu_inst"] if k else None


SIMDS = 1024            # MI355X: 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
FP32_VALU_PEAK = 157.3e12
CLOCK_HZ = 2.4e9        # shader clock (MI355X_MICROARCH.md)
SIMD_VALU_CYCLES = 2.0  # a wave64 fp32 VALU instruction occupies its SIMD-32 for 2 cycles (guide, per-instruction constants)
LONE_WAVE_CADENCE = 4.3 # cycles between two instructions of ONE wavefront, dependent or not (tools/micro/issue_cadence.hip)


def issue_view(sq, n, t_kernel=None, helper=False):
    """VALU-issue view of one launch from its SQ counters (quad-cycles summed over the launch's wavefronts).
    valu_issue_frac: busy fraction of the issue slots ONE wavefront can use (SQ_ACTIVE_INST_VALU per occupied SIMD / lifetime of
      a wave; 1.0 = an instruction every quad-cycle, the cadence of a lone wave -- which is HALF of what the SIMD can do).
    valu_simd_frac: fraction of the SIMD-32's VALU rate -- VALU instructions per occupied SIMD x 2 cycles / kernel cycles
      (t_kernel measured in this run x 2.4 GHz): the fraction of the MACHINE, the honest roof of a VALU-bound kernel.
    stall_frac: 1 - issue cycles of an environment's critical wavefront / kernel cycles, the issue cycles being its
      instructions of all kinds (VALU + SALU + LDS) x the 4.3-cycle cadence of a lone wave.  Helper-wave kernels: the main wave
      is charged with ALL of the environment's instructions (the helper's ~15 % included), so this is a LOWER bound of the
      stall share; kernels with W symmetric waves per environment: a W-th of them.
    None where a counter is missing."""
    try:
        waves = sq.get("SQ_WAVES") or None
        out = {"valu_insts_per_env_step": sq["SQ_INSTS_VALU"] / n, "lds_insts_per_env_step": sq["SQ_INSTS_LDS"] / n,
               "salu_insts_per_env_step": sq.get("SQ_INSTS_SALU", 0.0) / n}
        if waves:
            out["waves_per_env"] = waves / n
            # (a lone-wave view: undefined once the launch puts several rounds of waves on a SIMD -- the saturated regime is read
            # off valu_simd_frac)
            out["valu_issue_frac"] = ((sq["SQ_ACTIVE_INST_VALU"] / min(SIMDS, waves)) / (sq["SQ_WAVE_CYCLES"] / waves)
                                      if waves <= 2 * SIMDS else None)
            if t_kernel:
                cyc = t_kernel * CLOCK_HZ
                out["valu_simd_frac"] = sq["SQ_INSTS_VALU"] / min(SIMDS, waves) * SIMD_VALU_CYCLES / cyc
                per_env = (sq["SQ_INSTS_VALU"] + sq.get("SQ_INSTS_SALU", 0.0) + sq["SQ_INSTS_LDS"]) / n
                crit = per_env if helper else per_env / max(waves / n, 1.0)
                crit_waves = n if helper else waves
                # (defined while every critical wave has a SIMD to itself; with several per SIMD their waiting overlaps)
                out["stall_frac"] = 1.0 - crit * LONE_WAVE_CADENCE / cyc if crit_waves <= SIMDS else None
        if sq.get("SQ_LDS_BANK_CONFLICT") is not None and sq.get("SQ_ACTIVE_INST_LDS"):
            out["lds_bank_conflict_frac"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_ACTIVE_INST_LDS"]
        if sq.get("SQ_THREAD_CYCLES_VALU"):
            out["valu_active_lanes_avg"] = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"]
        return out
    except Exception:
        return None


def roofline_record(env, name, n, H, mm, device, reps, counters=True):
    """roofline object of one configuration: times the adjoint and the forward launch with HIP events (same shapes as the
    rollout), algorithmic bytes per SURVEY.md 8(d), and -- from the committed counter file of these very kernels -- measured HBM
    traffic, VALU instructions per env-step and the VALU-issue fraction that actually bounds these kernels."""
    t_bwd = time_backward_kernel(env, name, n, H, reps, device)
    t_fwd = time_backward_kernel.fwd_s
    nq, nd = env.num_joint_q, env.num_joint_qd
    na_in = env.model.muscles_per_articulation if env.model.muscle_count else nd
    # algorithmic bytes of ONE adjoint launch: re-read (q,qd,act) + read (gq',gqd') + write (gq,gqd,gact); one forward launch:
    # read (q,qd,act) + write (q',qd')  (the fused kernels also move obs / reward rows; SURVEY's figure is kept)
    bwd_bytes = 4 * n * ((nq + nd + na_in) + (nq + nd) + (nq + nd + na_in))
    fwd_bytes = 4 * n * ((nq + nd + na_in) + (nq + nd))
    # the roofline object is for the DOMINANT launch: the longer of the two kernels of an env-step (round 4: the adjoint; since
    # its body level became one phase the forward is as long or longer) -- `kernel` names it, `achieved` is ITS algorithmic bytes
    # over ITS duration; both kernels' times and byte counts are in the record either way
    fwd_dominant = t_fwd > t_bwd
    achieved = (fwd_bytes / t_fwd if fwd_dominant else bwd_bytes / t_bwd) / 1e9
    eng = env.model.engine()
    ckpt_floats = int(eng._lib.dsim_ckpt_floats_mm(eng._h, env.sim_substeps, mm))
    traffic = 4 * n * ckpt_floats + bwd_bytes
    traffic_src = "analytic: checkpoint words x N x 4 + boundary tensors (no counter file for these kernel sources)"
    pmc = pmc_record(name, n, mm) if counters else None   # (the committed counter files describe the SPECIALISED kernels)
    t_dom = t_fwd if fwd_dominant else t_bwd
    r = {"bound": "valu-issue", "kernel": "dsim_env_fwd_kernel" if fwd_dominant else "dsim_env_bwd_kernel",
         "dominant_launch": "forward" if fwd_dominant else "adjoint", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "alg_bytes_per_launch": fwd_bytes if fwd_dominant else bwd_bytes,
         # kernel_ms: the launch `kernel` names (the pair achieved / kernel_ms / traffic always describes ONE kernel);
         # both launches' own times are first-class next to it
         "kernel_ms": t_dom * 1e3, "fwd_kernel_ms": t_fwd * 1e3, "bwd_kernel_ms": t_bwd * 1e3,
         "fwd_alg_bytes_per_launch": fwd_bytes, "bwd_alg_bytes_per_launch": bwd_bytes,
         "fwd_alg_frac": fwd_bytes / t_fwd / (HBM_PEAK_GBS * 1e9), "bwd_alg_frac": bwd_bytes / t_bwd / (HBM_PEAK_GBS * 1e9),
         "csrc_hash": csrc_hash(),
         "ckpt_bytes_per_env_step": 4 * ckpt_floats, "ckpt_bytes_per_rollout": 4 * ckpt_floats * n * H,
         "valu_issue_frac": None, "fwd_valu_issue_frac": None, "valu_simd_frac": None, "fwd_valu_simd_frac": None,
         "stall_frac": None, "fwd_stall_frac": None, "valu_insts_per_env_step": None}
    if pmc:
        traffic, traffic_src = pmc["traffic_bytes_per_launch"], "rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE), " + pmc["file"]
        r["counters"] = pmc["file"]
        def helper_kernels(sq):   # one-wave mapping + helper wavefront: two waves per environment
            return bool(sq.get("SQ_WAVES")) and abs(sq["SQ_WAVES"] / n - 2.0) < 1e-6
        sa, sf = pmc.get("sq_adjoint") or {}, pmc.get("sq_forward") or {}
        b, f = issue_view(sa, n, t_bwd, helper_kernels(sa)), issue_view(sf, n, t_fwd, helper_kernels(sf))
        if b:
            r["valu_issue_frac"] = b.get("valu_issue_frac")
            r["valu_simd_frac"] = b.get("valu_simd_frac")
            r["stall_frac"] = b.get("stall_frac")
            r["adjoint"] = b
        if f:
            r["fwd_valu_issue_frac"] = f.get("valu_issue_frac")
            r["fwd_valu_simd_frac"] = f.get("valu_simd_frac")
            r["fwd_stall_frac"] = f.get("stall_frac")
            r["forward"] = f
        if b and f:
            r["valu_insts_per_env_step"] = b["valu_insts_per_env_step"] + f["valu_insts_per_env_step"]
            lanes = [x.get("valu_active_lanes_avg") for x in (b, f)]
            if all(lanes):
                # fp32 vector-ALU view: lane-operations actually executed (instructions x active lanes) ...
                ops_b, ops_f = n * b["valu_insts_per_env_step"] * lanes[0], n * f["valu_insts_per_env_step"] * lanes[1]
                # ... each counted as one FMA = 2 flop: an UPPER bound (moves, compares and selects are VALU instructions too)
                r["fp32_valu_frac_upper_bound"] = 2.0 * (ops_b + ops_f) / (t_bwd + t_fwd) / FP32_VALU_PEAK
                # ... and with the flops the instructions of these kernels actually carry (static census of the substep loops,
                # tools/opcode_census.py -> profiles/opcode_census.json: ~1.0 flop per VALU instruction -- a third are multiply-
                # adds, a third single operations, a third moves / selects / compares / integer work)
                cen = census_record()
                fb = census_flops_per_inst(cen, "dsim_env_bwd_kernel", name, b.get("waves_per_env"))
                ff = census_flops_per_inst(cen, "dsim_env_fwd_kernel", name, f.get("waves_per_env"))
                if fb and ff:
                    r["flops_per_valu_inst"] = {"adjoint": fb, "forward": ff, "source": "profiles/opcode_census.json"}
                    r["fp32_valu_frac_est"] = (ops_b * fb + ops_f * ff) / (t_bwd + t_fwd) / FP32_VALU_PEAK
                    r["flop_per_env_step"] = (ops_b * fb + ops_f * ff) / n
                else:
                    r["fp32_valu_frac_est"] = None
    # measured HBM traffic per launch, each kernel against ITS OWN duration; `traffic` / `hbm_measured_frac` are the dominant
    # launch's pair (like `achieved`), the other kernel's stay in bwd_* / fwd_*
    wf = (pmc.get("forward_kernel") or {}).get("traffic_bytes_per_launch") if pmc else None
    r["bwd_traffic"], r["fwd_traffic"], r["traffic_source"] = traffic, wf, traffic_src
    r["bwd_hbm_measured_frac"] = traffic / t_bwd / (HBM_PEAK_GBS * 1e9)
    r["fwd_hbm_measured_frac"] = wf / t_fwd / (HBM_PEAK_GBS * 1e9) if wf else None
    if fwd_dominant:
        # (no counter file: the forward writes the checkpoint block the adjoint reads back -- the analytic figure minus the
        # gradient tensors)
        r["traffic"] = wf if wf else 4 * n * ckpt_floats + fwd_bytes
        r["hbm_measured_frac"] = r["traffic"] / t_fwd / (HBM_PEAK_GBS * 1e9)
    else:
        r["traffic"], r["hbm_measured_frac"] = traffic, r["bwd_hbm_measured_frac"]
    return r


def measure_other_config(name, n, H, mm, device, steps=10, generic=False):
    """fwd+adjoint env-steps/s of another BASELINE.json configuration on this GPU, measured like the headline: the same
    rollout (H x DFlexEnv.step, loss = -sum(rew), one backward) captured as a HIP graph, `steps` timed replays after two
    warm-up replays, and its own roofline object (kernel times by HIP events, counters from the matching profiles/ file)."""
    saved = MM_FREQ[name]
    MM_FREQ[name] = mm
    if generic:
        os.environ["DSIM_FORCE_GENERIC"] = "1"   # read by dsim_model_create (diffrl_amd/csrc/dsim_hip.hip: match_variant)
    try:
        from diffrl_amd.graph import GraphedRollout
        env = make_env(name, n, str(device))
        assert (env.model.engine().variant == 0) == bool(generic)
        gen = torch.Generator().manual_seed(1)
        actions = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0).to(device)
        acts = actions.detach().clone().requires_grad_(True)

        def body(e):
            e.initialize_trajectory()
            return reward_loss([e.step(a_t)[1] for a_t in acts.unbind(0)])

        env.clear_grad()
        env.reset()
        roll = GraphedRollout(env, body, leaves=[acts], carry_state=False)
        for _ in range(2):
            roll.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            roll.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert torch.isfinite(acts.grad).all()
        rf = roofline_record(env, name, n, H, mm, device, 20, counters=not generic)
        rf["alg_frac_step"] = ALG_BYTES[name] * (steps * n * H / el) / (HBM_PEAK_GBS * 1e9)
        return {"workload": "%s %d envs x H=%d, MM_caching_frequency %d" % (name, n, H, mm), "value": steps * n * H / el,
                "unit": "env-steps/s", "steps": steps, "ms_per_rollout": el / steps * 1e3, "kernel_ms": rf["kernel_ms"],
                "fwd_kernel_ms": rf["fwd_kernel_ms"], "bwd_kernel_ms": rf["bwd_kernel_ms"],
                "ckpt_bytes_per_env_step": rf["ckpt_bytes_per_env_step"],
                "kernels": "generic (run-time layout)" if generic else "specialised (compile-time layout)", "roofline": rf}
    except Exception as ex:
        return {"workload": "%s %d envs x H=%d" % (name, n, H), "value": None, "error": str(ex)[:200]}
    finally:
        MM_FREQ[name] = saved
        if generic:
            os.environ.pop("DSIM_FORCE_GENERIC", None)


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


MAX_LINE_BYTES = 4096   # the driver reads the LAST stdout line out of a bounded tail: round 5's 21 KB line was not parsed


def _sig(x, digits=5):
    """floats to `digits` significant digits (the printed line only; the full record keeps every bit)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "fwd_kernel_ms", "bwd_kernel_ms",
                 "fwd_alg_frac", "bwd_alg_frac", "alg_frac_step", "alg_bytes_per_launch", "hbm_measured_frac",
                 "fwd_hbm_measured_frac", "bwd_hbm_measured_frac", "valu_simd_frac", "fwd_valu_simd_frac", "fp32_valu_frac_est",
                 "counters", "csrc_hash")


def compact_line(full, side_file=None):
    """the ONE line bench.py prints: the contract's keys, a numeric-only roofline, the CPU baseline's figures and one short
    object per other configuration -- under MAX_LINE_BYTES whatever the number of ranks.  Everything else (per-kernel counter
    views, samples, sources, per-config roofline objects) is in the full record (`side_file`)."""
    keep = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "backend", "oversubscribed", "steps", "warmup", "ms_per_step",
            "ms_per_step_min", "ms_per_step_median", "ms_per_step_max", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "eager_env_steps_per_s", "no_grad_forward_env_steps_per_s", "dry_run", "envs_total", "slowest_rank_s")
    out = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get("config") or {})
    if "submission" in cfg:
        cfg["submission"] = "eager" if cfg["submission"].startswith("eager") else "hip-graph"
    out["config"] = cfg
    if full.get("per_rank"):
        out["per_rank"] = [{k: r[k] for k in ("rank", "value", "ms_per_step", "ms_per_step_median", "ms_per_step_max", "elapsed_s",
                                              "envs", "device") if k in r} for r in full["per_rank"]]
    rf = full.get("roofline")
    if rf:
        out["roofline"] = {k: rf.get(k) for k in ROOFLINE_KEYS}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                               "sample": (cb.get("sample") or "")[:120],
                               "single_thread": (cb.get("single_thread") or {}).get("value"),
                               "reference_recorded": (cb.get("reference_recorded") or {}).get("value"),
                               "reference_unavailable": (cb.get("reference_unavailable") or None) and cb["reference_unavailable"][:160]}
    if full.get("other_configs"):
        oc = []
        for o in full["other_configs"]:
            r = o.get("roofline") or {}
            c = {"workload": o["workload"].replace(", MM_caching_frequency ", " mm") + (" generic" if str(o.get("kernels", "")).startswith("generic") else ""),
                 "value": o.get("value"), "fwd_ms": o.get("fwd_kernel_ms"), "bwd_ms": o.get("bwd_kernel_ms"),
                 "frac": r.get("frac"), "alg_frac_step": r.get("alg_frac_step"), "hbm_measured_frac": r.get("hbm_measured_frac")}
            if o.get("error"):
                c["error"] = o["error"][:80]
            oc.append(c)
        out["other_configs"] = oc
    if full.get("note"):
        out["note"] = full["note"][:200]
    if side_file:
        out["full_record"] = side_file
    line = json.dumps(_sig(out), separators=(",", ":"))
    if len(line) >= MAX_LINE_BYTES:   # (cannot happen with <= 8 ranks; keep the contract's keys whatever grows)
        for k in ("per_rank", "other_configs", "note"):
            out.pop(k, None)
        line = json.dumps(_sig(out), separators=(",", ":"))
    assert len(line) < MAX_LINE_BYTES, len(line)
    return line


def emit(full):
    """writes the full record next to bench.py (bench_full.json; gpurun_out/ too when it exists) and prints the compact line"""
    side = None
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    json.dump(full, f, indent=1)
                side = side or os.path.relpath(os.path.join(d, "bench_full.json"), ROOT)
            except OSError:
                pass
    sys.stdout.flush()
    print(compact_line(full, side), flush=True)


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
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the informational legs (Python-driven eager loop, forward-only no-grad loop): under rocprofv3 every "
                         "launch of a kernel is then a launch of the timed rollout or of the kernel timing (tools/profile.sh)")
    ap.add_argument("--eager", action="store_true", help="time the Python-driven step loop instead of the graph replay")
    ap.add_argument("--strict", action="store_true", help="exit non-zero if the graph capture fell back to the eager loop")
    ap.add_argument("--launcher", action="store_true",
                    help="go through torch.distributed.run even for --gpus 1 (an RCCL group of one rank)")
    ap.add_argument("--oversubscribe", type=int, default=1,
                    help="R > 1: R ranks PER GPU (backend gloo; RCCL refuses two ranks on one device).  NOT a scaling measurement: "
                         "it exists so that the launcher -> shard -> Engine -> timing all-reduce path executes with world_size > 1 "
                         "against real kernels on a 1-GPU box; the line says oversubscribed: true and keeps n_gpus = --gpus")
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
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus * a.oversubscribe),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    if a.gpus < 1 or a.oversubscribe < 1:
        sys.stderr.write("bench.py: --gpus and --oversubscribe must be >= 1\n")
        return 2
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and (a.gpus > 1 or a.oversubscribe > 1 or a.launcher or a.dry_run):
        return launch_ranks(a, argv)

    from diffrl_amd import sharding
    rank, local, world = sharding.world()
    over = a.oversubscribe > 1
    if world != a.gpus * a.oversubscribe:
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d%s; the label must match the job\n"
                         % (world, a.gpus, (" x --oversubscribe %d" % a.oversubscribe) if over else ""))
        return 2
    local = local // a.oversubscribe   # R consecutive local ranks share a device
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
        # (RCCL needs one device per rank: oversubscribed ranks talk over gloo, with host tensors)
        sharding.init("nccl" if (use_gpu and not over) else "gloo", device if (use_gpu and not over) else None)
        rccl_ranks = td.get_world_size()
        if rccl_ranks != a.gpus * a.oversubscribe:   # (never print a job of another size under the requested label)
            sys.stderr.write("bench.py: the process group has %d ranks, --gpus %d x --oversubscribe %d asked for %d\n"
                             % (rccl_ranks, a.gpus, a.oversubscribe, a.gpus * a.oversubscribe))
            return 2
    red_dev = device if (use_gpu and not over) else torch.device("cpu")   # where the timing all-reduce lives

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
        el_rank = time.perf_counter() - t0   # this rank's own work, before it waits for the others
        barrier()
        el = sharding.max_over_ranks(time.perf_counter() - t0, red_dev)
        owned = sharding.sum_over_ranks(hi - lo, red_dev)
        per_rank = sharding.gather_over_ranks([rank, el_rank, hi - lo], red_dev)
        names = sharding.gather_strings(torch.cuda.get_device_name(device) if use_gpu else "cpu")
        if rank == 0:
            print(compact_line({"metric": "fwd+adjoint env-steps/sec", "dry_run": True, "value": None, "unit": "env-steps/s",
                              "per_rank": [{"rank": int(r[0]), "elapsed_s": r[1], "envs": int(r[2]), "device": names[i]}
                                           for i, r in enumerate(per_rank)],
                              "n_gpus": a.gpus, "rccl_ranks": rccl_ranks, "backend": "nccl" if (use_gpu and not over) else "gloo",
                              "oversubscribed": over,
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
    # one HIP event per timed step on the launch stream (the replays and the eager launches go to torch's current stream):
    # min / median / max per step next to the wall-clock figure, so that a sub-percent change can be told from noise
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if use_gpu else []
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        if marks:
            marks[i].record()
        grad = one()
    if marks:
        marks[a.steps].record()
    if use_gpu:
        torch.cuda.synchronize()
    el_own = time.perf_counter() - t0   # this rank's own K steps, before it waits for the others (per_rank in the JSON line)
    barrier()
    el = time.perf_counter() - t0
    gc.enable()
    per_step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)) if marks else []
    assert torch.isfinite(grad).all()
    el_rank = el_own
    el = sharding.max_over_ranks(el, red_dev)
    total_env_steps = a.steps * world * n * H
    value = total_env_steps / el
    # every rank's own figures on rank 0 (straggler diagnosis of the 2 / 4 / 8-GPU runs: the job's time is the slowest rank's)
    per_rank = sharding.gather_over_ranks([rank, el_rank, per_step_ms[len(per_step_ms) // 2] if per_step_ms else 0.0,
                                           per_step_ms[-1] if per_step_ms else 0.0], red_dev)
    dev_names = sharding.gather_strings("%s (cuda:%d)" % (torch.cuda.get_device_name(device), device.index) if use_gpu else "cpu")
    eager_value = None
    # (the informational legs below are rank 0's alone: with several ranks they are skipped, so that no rank waits more than the
    # second or two of the kernel timings at the closing barrier)
    if roll is not None and rank == 0 and not a.no_extras and world == 1:
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
        rf = roofline_record(env, a.env, n, H, mm, device, 50)
        eng = env.model.engine()
        out = {
            "metric": "fwd+adjoint env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": a.gpus,
            "rccl_ranks": rccl_ranks, "oversubscribed": over,
            "backend": ("nccl" if (use_gpu and not over) else "gloo") if dist else None,
            "per_rank": [{"rank": int(r[0]), "value": a.steps * n * H / r[1], "ms_per_step": r[1] / a.steps * 1e3,
                          "ms_per_step_median": r[2], "ms_per_step_max": r[3], "device": dev_names[i]}
                         for i, r in enumerate(per_rank)],
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "ms_per_step_min": per_step_ms[0] if per_step_ms else None,
            "ms_per_step_median": per_step_ms[len(per_step_ms) // 2] if per_step_ms else None,
            "ms_per_step_max": per_step_ms[-1] if per_step_ms else None,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %d envs/GPU x H=%d through DFlexEnv.step, loss=-sum(rew), 1 backward"
                                   % (a.env, n, H), "envs_per_gpu": n, "envs_total": n * world, "horizon": H,
                       "substeps": env.sim_substeps,
                       "mm_freq": mm, "sharding": "envs by index, no collective", "submission": submission,
                       "submission_fallback": fallback},
            "eager_env_steps_per_s": eager_value,
            "roofline": rf,
            "fp32_valu_frac_est": rf.get("fp32_valu_frac_est"),
        }
        rf["alg_frac_step"] = ALG_BYTES[a.env] * (value / world) / (HBM_PEAK_GBS * 1e9)   # per GPU: whole env-step bytes x rate / 8 TB/s
        # forward-only serving path (dflex.config.no_grad: no checkpoint traffic), SURVEY.md 8(f).4 -- informational
        if not a.no_extras and world == 1:
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
        if over:
            out["backend"] = "gloo"
            out["note"] = ("%d ranks share each GPU (--oversubscribe): the multi-rank launcher / sharding / timing path executed against "
                           "real kernels, NOT a multi-GPU figure -- the ranks' launches time-slice one device" % a.oversubscribe)
        if not a.no_other_configs and world == 1 and a.env == "ant" and not a.eager:
            # BASELINE.json configs[2], configs[3] and the MM_caching_frequency = 1 variant of configs[1] (SURVEY.md 8(d):
            # "also report 1"), each a few seconds: driver-visible numbers next to the headline
            del env, roll
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = [measure_other_config("humanoid", 1024, H, MM_FREQ["humanoid"], device),
                                    measure_other_config("snu", 512, H, MM_FREQ["snu"], device),
                                    measure_other_config("ant", n, H, 1, device),
                                    # beyond the helper-wave capacity: forward with two environments per wavefront
                                    # (dsim_hip.hip: DSIM_MODE_PAIR), adjoint with one
                                    measure_other_config("ant", 8192, H, MM_FREQ["ant"], device, steps=5),
                                    # the headline workload on the GENERIC kernels (run-time layout: what a user model that
                                    # matches no specialised kernel set gets; tools/gen_static_layouts.py adds a set)
                                    measure_other_config("ant", n, H, MM_FREQ["ant"], device, generic=True)]
        if not a.no_cpu_baseline and world == 1:   # reported at N = 1 only (the host cores are the same for every N)
            out["cpu_baseline"] = cpu_baseline(a.env)
        emit(out)
    if dist:
        td.barrier()   # rank 0 is still measuring its extras (eager loop, CPU baseline): tear the group down together
        td.destroy_process_group()
    if fallback and a.strict:
        sys.stderr.write("bench.py: --strict and the graph capture fell back to the eager loop\n")
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
