"""TEST/BENCH INFRASTRUCTURE: the CPU baseline leg of bench.py (run as a subprocess so that no GPU runtime state is
forked).  Prints one JSON object.

Two legs:
  * kind "reference": when the reference checkout is present (build container: /root/reference or $DIFFRL_REFERENCE),
    the REAL reference (NVlabs/DiffRL dflex CPU path, generated kernels of dflex/dflex/adjoint.py:1271-1289, which are
    single-threaded by construction) is run through oracle/ref_harness.py on a bounded sample of the bench workload:
    the same env class / substeps / MM_caching_frequency / seeded tanh actions / loss = -sum(reward) / one backward,
    at a reduced size (envs x horizon below) so that it finishes in ~10-30 s.
  * kind "port": the scalar restatement oracle/dsim_oracle.cpp (forward + taped reverse sweep of one env.step, the
    operator the benchmark's kernels implement), over T host threads inside ONE process (environments are independent:
    contiguous env ranges, one tape per thread; dsim_oracle_step_backward_mt), T stated in `cores`.
The GPU box has no reference checkout: there the port runs and the recorded reference figure
is attached as `reference_recorded`.

usage: cpu_baseline.py <env> <budget seconds> [threads]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

# the survey's measurement of the reference's own CPU path in the build container
REFERENCE_RECORDED = {"ant": {"value": 696.5, "unit": "env-steps/s", "cores": 8, "kernel_threads": 1,
                              "source": "BASELINE.md section 2: reference CPU path, Ant 1024 envs x H=32 fwd+bwd, 8 vCPU container",
                              # this script's own reference leg in the build container, round 6 (bounded sample 256 envs x H=8)
                              "round6_sample": {"value": 1036.3, "file": "profiles/r06_cpu_baseline_reference.json"}}}
REF_SAMPLE = {"ant": (256, 8), "humanoid": (32, 4), "snu": (16, 4), "cartpole": (64, 16), "hopper": (256, 8), "cheetah": (256, 8)}
MM_FREQ = {"ant": 16, "humanoid": 48, "snu": 8, "cartpole": 4, "hopper": 16, "cheetah": 16}


def ref_root():
    return os.environ.get("DIFFRL_REFERENCE", "/root/reference")


def port_leg(name, seconds, threads):
    import ctypes as C

    import numpy as np
    from oracle_lib import _c, _p, golden, oracle, template_from_golden

    from diffrl_amd.capi import make_desc

    t = template_from_golden(name)
    g = golden(name + "_step")
    S, mm, dt = int(g["substeps"]), int(g["mm_freq"]), float(g["dt"])
    per_thread = 8
    n = threads * per_thread
    reps = n // g["q_in"].shape[0] + 1
    tile = lambda a: _c(np.tile(a, (reps, 1))[:n])
    q, qd, act, gq_o, gqd_o = tile(g["q_in"]), tile(g["qd_in"]), tile(g["act_in"]), tile(g["gq_out"]), tile(g["gqd_out"])
    mact = tile(g["muscle_act_in"]) if "muscle_act_in" in g else np.zeros((n, 0), np.float32)
    gq, gqd, ga, gm = np.zeros_like(q), np.zeros_like(qd), np.zeros_like(act), np.zeros_like(mact)
    desc, keep = make_desc(t)
    fn = oracle().dsim_oracle_step_backward_mt
    fn.restype = C.c_int

    def call():
        rc = fn(C.byref(desc), C.c_int(n), C.c_int(threads), _p(q), _p(qd), _p(act), _p(mact), C.c_float(dt), C.c_int(S),
                C.c_int(mm), _p(gq_o), _p(gqd_o), _p(gq), _p(gqd), _p(ga), _p(gm), None, None)
        assert rc == 0

    call()
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        call()
        done += n
    el = time.perf_counter() - t0
    assert np.isfinite(gq).all() and np.isfinite(ga).all()
    return {"value": done / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d %s env-steps (forward + taped reverse sweep, oracle/dsim_oracle.cpp) in %.1f s on %d host threads "
                      "of %d cores, environments split over the threads" % (done, name, el, threads, os.cpu_count())}


def reference_leg(name, seconds):
    """the reference itself, through its own DFlexEnv loop (what bench.py's rollout() does on the GPU)"""
    import numpy as np
    import torch

    import ref_harness
    from gen_golden import CONFIGS
    df, envs = ref_harness.load_reference()
    n, H = REF_SAMPLE[name]
    cls, _, _, _, has_et = CONFIGS[name]
    kw = dict(num_envs=n, device="cpu", render=False, seed=0, episode_length=100000, no_grad=False,
              stochastic_init=False, MM_caching_frequency=MM_FREQ[name])
    if has_et:
        kw["early_termination"] = False
    torch.manual_seed(0)
    np.random.seed(0)
    env = getattr(envs, cls)(**kw)
    gen = torch.Generator().manual_seed(1)
    actions = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0)

    def rollout():
        env.clear_grad()
        env.reset()
        env.initialize_trajectory()
        acts = actions.detach().requires_grad_(True)
        loss = 0.0
        for a_t in acts.unbind(0):
            obs, rew, done, info = env.step(a_t)
            loss = loss - rew.sum()
        loss.backward()
        return acts.grad

    rollout()   # warm-up (kernel .so already JIT-built by load_reference)
    done, t0 = 0, time.perf_counter()
    while True:
        g = rollout()
        done += n * H
        if time.perf_counter() - t0 >= seconds:
            break
    el = time.perf_counter() - t0
    assert torch.isfinite(g).all()
    return {"value": done / el, "unit": "env-steps/s", "cores": 1, "kind": "reference",
            "sample": "%d %s env-steps: the reference's own CPU path (dflex generated kernels, single kernel thread; "
                      "torch intra-op threads %d) on %d envs x H=%d rollouts (fwd + backward of -sum(rew)), %.1f s, "
                      "host has %d cores" % (done, name, torch.get_num_threads(), n, H, el, os.cpu_count())}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ant"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else min(os.cpu_count() or 1, 64)
    out = None
    if os.environ.get("DSIM_CPU_BASELINE", "auto") != "port":
        try:
            import ref_harness
            if ref_harness.reference_available():
                out = reference_leg(name, seconds)
        except Exception as ex:  # reference present but not runnable here: say so and fall back to the port
            out = None
            note = "reference leg failed: %s" % str(ex)[:160]
        else:
            note = None
    else:
        note = None
    if out is None:
        out = port_leg(name, seconds, threads)
        # the port on ONE thread next to the multi-threaded figure (its taped reverse sweep allocates
        # heavily and does not scale linearly with threads: the single-thread figure is the comparable per-core number)
        if threads > 1:
            st = port_leg(name, max(2.0, seconds / 4), 1)
            out["single_thread"] = {k: st[k] for k in ("value", "unit", "cores", "sample")}
        if note:
            out["note"] = note
        if name in REFERENCE_RECORDED:
            out["reference_recorded"] = REFERENCE_RECORDED[name]
        # why kind is "port" on this box: the reference is a Python package (its kernels are C++ that its own
        # code generator writes from Python functions at import time, dflex/dflex/adjoint.py:1749-1898); a Python reference may not
        # travel to the GPU box in any form -- source, bytecode or the .so its generator builds -- and /root/reference does not
        # exist there.  Where the checkout IS present (the build container) this script times the reference itself (kind
        # "reference"; profiles/r06_cpu_baseline_reference.json is that run).
        out["reference_unavailable"] = note or ("no reference checkout on this box (%s): a Python reference cannot travel; its own CPU "
                                                "path is timed where the checkout exists (profiles/r06_cpu_baseline_reference.json)"
                                                % ref_root())
    else:
        # the port's figure next to the reference's, same run (half the budget)
        p = port_leg(name, max(2.0, seconds / 2), threads)
        out["port"] = {k: p[k] for k in ("value", "unit", "cores", "sample")}
        if threads > 1:
            st = port_leg(name, max(2.0, seconds / 4), 1)
            out["port"]["single_thread"] = {k: st[k] for k in ("value", "unit", "cores", "sample")}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
