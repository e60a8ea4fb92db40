"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the REAL
reference (CPU path of NVlabs/DiffRL dflex, /root/reference) in this container.

The reference ships no golden vectors / known-answer tests for the articulated
path, so parity is pinned by these files instead:

  <env>_model.npz    single-articulation model constants as the reference's
                     ModelBuilder/finalize/collide produced them (env 0 slice)
                     -> pins diffrl_amd's own asset parsers + model build
  <env>_step.npz     operator level: SemiImplicitIntegrator.forward
                     (dflex/dflex/sim.py:2182) on B states x (q, qd, act) ->
                     (q', qd') and the adjoint for random cotangents; plus every
                     intermediate tensor of the first substep
  <env>_rollout.npz  DFlexEnv level: H env.step() calls + backward of
                     -sum(rew) w.r.t. the actions (envs/<env>.py)
  ant_rollout_h32.npz  the same at BASELINE.json's horizon (H = 32, 8 envs)
  ant_1024x32.npz    BASELINE.json configs[1] literally: 1024 envs, H = 32 (slimmed, see main; `... ant_1024x32`)
  cartpole_rollout_64x16.npz   BASELINE.json configs[0] literally: 64 envs, H = 16 (`... cartpole_64x16`)
  humanoid_rollout_h32.npz, snu_rollout_h32.npz   H = 32, 2 envs (`python oracle/gen_golden.py h32_extra`)
  <env>_rollout_mm1.npz  (`... mm1`) Ant 8 x H=8, SNU 2 x H=4, Humanoid 2 x H=4 with MM_caching_frequency = 1
  humanoid_exploded.npz  (`... exploded`) a rollout in which two environments blow up (finite > 1e6, and inf / NaN)
  <env>_episode.npz  (`... ant_extra`, `... episodes_extra`)  H steps WITH the reference's episode handling active:
                     early termination on, episode_length = 10, so every env is
                     reset (envs/ant.py:176-234) at least twice inside the
                     rollout; records done / obs_before_reset per step and the
                     gradient of a loss that also reads obs_before_reset
                     (`python oracle/gen_golden.py ant_extra`)

Usage:  python oracle/gen_golden.py [ant humanoid snu cartpole hopper cheetah]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

CONFIGS = {
    # name: (class, mm_freq (examples/cfg/shac/*.yaml), n_envs, rollout H, has early_termination kw)
    "cartpole": ("CartPoleSwingUpEnv", 4, 4, 6, True),
    "ant": ("AntEnv", 16, 4, 6, True),
    "humanoid": ("HumanoidEnv", 48, 3, 3, False),
    "snu": ("SNUHumanoidEnv", 8, 3, 3, False),
    "hopper": ("HopperEnv", 16, 4, 6, True),
    "cheetah": ("CheetahEnv", 16, 4, 6, True),
}


def t2n(t):
    return t.detach().cpu().numpy().copy()


def make_env(envs, name, n, no_grad=False, stochastic=True, mm=None):
    cls, mmf, _, _, has_et = CONFIGS[name]
    kw = dict(num_envs=n, device="cpu", render=False, seed=0, episode_length=1000, no_grad=no_grad,
              stochastic_init=stochastic, MM_caching_frequency=mm if mm is not None else mmf)
    if has_et:
        kw["early_termination"] = False
    torch.manual_seed(0)
    np.random.seed(0)
    return getattr(envs, cls)(**kw)


def dump_model(env, n):
    m = env.model
    L = m.link_count // n
    nq = m.joint_coord_count // n
    nd = m.joint_dof_count // n
    d = dict(
        n_links=L, n_q=nq, n_qd=nd,
        joint_type=t2n(m.joint_type[:L]), joint_parent=t2n(m.joint_parent[:L]),
        joint_q_start=t2n(m.joint_q_start[:L + 1]), joint_qd_start=t2n(m.joint_qd_start[:L + 1]),
        joint_X_pj=t2n(m.joint_X_pj[:L]), joint_X_cm=t2n(m.joint_X_cm[:L]), joint_axis=t2n(m.joint_axis[:L]),
        body_I_m=t2n(m.body_I_m[:L]), joint_armature=t2n(m.joint_armature[:nd]),
        joint_target=t2n(m.joint_target[:nq]), joint_target_ke=t2n(m.joint_target_ke[:L]),
        joint_target_kd=t2n(m.joint_target_kd[:L]), joint_limit_lower=t2n(m.joint_limit_lower[:nq]),
        joint_limit_upper=t2n(m.joint_limit_upper[:nq]), joint_limit_ke=t2n(m.joint_limit_ke[:L]),
        joint_limit_kd=t2n(m.joint_limit_kd[:L]), joint_q0=t2n(m.joint_q[:nq]), joint_qd0=t2n(m.joint_qd[:nd]),
        gravity=t2n(m.gravity), ground=int(bool(m.ground)),
        dt=np.float64(env.sim_dt), substeps=env.sim_substeps,
    )
    ns = m.shape_count // n
    d["shape_materials"] = t2n(m.shape_materials[:ns]).reshape(ns, 4)
    d["shape_body"] = t2n(m.shape_body[:ns])
    d["shape_geo_type"] = t2n(m.shape_geo_type[:ns])
    d["shape_geo_scale"] = t2n(m.shape_geo_scale[:ns]).reshape(ns, 3)
    d["shape_transform"] = t2n(m.shape_transform[:ns]).reshape(ns, 7)
    if m.ground and m.contact_count > 0:
        C = m.contact_count // n
        d.update(contact_body=t2n(m.contact_body0[:C]), contact_point=t2n(m.contact_point0[:C]),
                 contact_dist=t2n(m.contact_dist[:C]), contact_material=t2n(m.contact_material[:C]))
    else:
        d.update(contact_body=np.zeros(0, np.int32), contact_point=np.zeros((0, 3), np.float32),
                 contact_dist=np.zeros(0, np.float32), contact_material=np.zeros(0, np.int32))
    M = m.muscle_count // n
    if M > 0:
        W = int(m.muscle_start[M].item())
        d.update(muscle_start=t2n(m.muscle_start[:M + 1]), muscle_links=t2n(m.muscle_links[:W]),
                 muscle_points=t2n(m.muscle_points[:W]), muscle_params=t2n(m.muscle_params[:M]))
        d["muscle_strengths"] = t2n(env.muscle_strengths[:M])
    else:
        d.update(muscle_start=np.zeros(1, np.int32), muscle_links=np.zeros(0, np.int32),
                 muscle_points=np.zeros((0, 3), np.float32))
    # replication sanity: env 1 must be env 0 shifted by the link offset
    if n > 1:
        assert torch.equal(m.joint_parent[L:2 * L] - L * (m.joint_parent[L:2 * L] >= 0).int(), m.joint_parent[:L])
        assert torch.equal(m.body_I_m[L:2 * L], m.body_I_m[:L])
        assert torch.equal(m.joint_X_pj[L + 1:2 * L], m.joint_X_pj[1:L])
    return d


def collect_states(envs, name, n, steps_at):
    """Roll the no-grad reference env with random actions, return (q, qd) snapshots."""
    env = make_env(envs, name, n, no_grad=True)
    env.reset()
    g = torch.Generator().manual_seed(7)
    snaps = []
    for t in range(max(steps_at) + 1):
        if t in steps_at:
            snaps.append((t2n(env.state.joint_q).reshape(n, -1), t2n(env.state.joint_qd).reshape(n, -1)))
        a = torch.tanh(2.0 * torch.rand((n, env.num_actions), generator=g) - 1.0) * 1.0
        env.step(a)
    q = np.concatenate([s[0] for s in snaps], 0)
    qd = np.concatenate([s[1] for s in snaps], 0)
    return q, qd


def step_golden(df, envs, name):
    cls, mmf, n, _, _ = CONFIGS[name]
    steps_at = {"cartpole": [0, 5], "ant": [0, 12, 30], "humanoid": [0, 10], "snu": [0, 10],
                "hopper": [0, 20], "cheetah": [0, 20]}[name]
    q, qd = collect_states(envs, name, n, steps_at)
    B = q.shape[0]
    env = make_env(envs, name, B, no_grad=False, stochastic=False)
    df.config.no_grad = False
    model, integ = env.model, env.integrator
    nd = model.joint_dof_count // B
    g = torch.Generator().manual_seed(11)
    state = model.state()
    state.joint_q = torch.tensor(q.reshape(-1), dtype=torch.float32, requires_grad=True)
    state.joint_qd = torch.tensor(qd.reshape(-1), dtype=torch.float32, requires_grad=True)
    out = {"q_in": q, "qd_in": qd, "mm_freq": mmf, "substeps": env.sim_substeps, "dt": np.float64(env.sim_dt)}
    M = model.muscle_count // B
    if M > 0:
        act = torch.rand((B, M), generator=g) * torch.tensor(t2n(env.muscle_strengths)).view(B, M)
        act = act.clone().requires_grad_(True)
        model.muscle_activation = act.view(-1)
        out["muscle_act_in"] = t2n(act)
        jact = torch.zeros(B * nd)
        state.joint_act = jact
        out["act_in"] = np.zeros((B, nd), np.float32)
    else:
        scale = {"cartpole": 1000.0, "ant": 200.0, "humanoid": 60.0, "hopper": 200.0, "cheetah": 200.0}[name]
        jact = ((2.0 * torch.rand((B, nd), generator=g) - 1.0) * scale)
        if int(model.joint_type[0]) == 4:
            jact[:, :6] = 0.0
        jact = jact.clone().requires_grad_(True)
        state.joint_act = jact.view(-1)
        out["act_in"] = t2n(jact)

    # --- every intermediate of the FIRST substep (sim.py:2225-2601), on a throw-away tape
    s1 = model.state()
    s_in = model.state()
    s_in.joint_q = state.joint_q.detach().clone()
    s_in.joint_qd = state.joint_qd.detach().clone()
    s_in.joint_act = state.joint_act.detach().clone()
    h = env.sim_dt / float(env.sim_substeps)
    integ._simulate(df.adjoint.Tape(), model, s_in, s1, h, update_mass_matrix=True)
    L = model.link_count // B
    out.update(sub_X_sc=t2n(s1.body_X_sc).reshape(B, L, 7), sub_X_sm=t2n(s1.body_X_sm).reshape(B, L, 7),
               sub_S_s=t2n(s1.joint_S_s).reshape(B, nd, 6), sub_I_s=t2n(s1.body_I_s).reshape(B, L, 6, 6),
               sub_v_s=t2n(s1.body_v_s).reshape(B, L, 6), sub_a_s=t2n(s1.body_a_s).reshape(B, L, 6),
               sub_f_s=t2n(s1.body_f_s).reshape(B, L, 6), sub_ft_s=t2n(s1.body_ft_s).reshape(B, L, 6),
               sub_tau=t2n(s1.joint_tau).reshape(B, nd), sub_qdd=t2n(s1.joint_qdd).reshape(B, nd),
               sub_H=t2n(model.H).reshape(B, nd, nd), sub_L=t2n(model.L).reshape(B, nd, nd),
               sub_q=t2n(s1.joint_q).reshape(B, -1), sub_qd=t2n(s1.joint_qd).reshape(B, -1), sub_dt=np.float64(h))

    # --- full env-step through the autograd op
    so = integ.forward(model, state, env.sim_dt, env.sim_substeps, mmf)
    gq = torch.randn(so.joint_q.shape, generator=g)
    gqd = torch.randn(so.joint_qd.shape, generator=g)
    loss = (so.joint_q * gq).sum() + (so.joint_qd * gqd).sum()
    loss.backward()
    out.update(q_out=t2n(so.joint_q).reshape(B, -1), qd_out=t2n(so.joint_qd).reshape(B, -1),
               gq_out=t2n(gq).reshape(B, -1), gqd_out=t2n(gqd).reshape(B, -1),
               gq_in=t2n(state.joint_q.grad).reshape(B, -1), gqd_in=t2n(state.joint_qd.grad).reshape(B, -1))
    if M > 0:
        out["gmuscle_act_in"] = t2n(act.grad)
    else:
        out["gact_in"] = t2n(jact.grad)
    return out


def rollout_golden(df, envs, name):
    cls, mmf, n, H, _ = CONFIGS[name]
    env = make_env(envs, name, n, no_grad=False)
    env.clear_grad()
    env.reset()
    g = torch.Generator().manual_seed(3)
    # pre-roll so that the recorded horizon has active ground contacts
    for t in range({"ant": 20, "humanoid": 8, "snu": 8, "hopper": 20, "cheetah": 20}.get(name, 0)):
        with torch.no_grad():
            _, _, done, _ = env.step(torch.tanh(2.0 * torch.rand((n, env.num_actions), generator=g) - 1.0))
        assert int(done.sum()) == 0
        env.clear_grad()
    q0, qd0 = env.get_state()
    obs0 = env.initialize_trajectory()
    acts = (2.0 * torch.rand((H, n, env.num_actions), generator=g) - 1.0)
    acts = torch.tanh(acts).clone().requires_grad_(True)
    obs_l, rew_l = [], []
    loss = 0.0
    for t in range(H):
        obs, rew, done, info = env.step(acts[t])
        assert int(done.sum()) == 0, "golden rollouts must not auto-reset"
        obs_l.append(t2n(obs))
        rew_l.append(t2n(rew))
        loss = loss - rew.sum()
    loss.backward()
    return dict(q0=t2n(q0).reshape(n, -1), qd0=t2n(qd0).reshape(n, -1), obs0=t2n(obs0), actions=t2n(acts),
                obs=np.stack(obs_l), rew=np.stack(rew_l), grad_actions=t2n(acts.grad), mm_freq=mmf,
                q_final=t2n(env.state.joint_q).reshape(n, -1), qd_final=t2n(env.state.joint_qd).reshape(n, -1),
                loss=np.float64(loss.item()))


def episode_golden(envs, name, n, H, L, act_gain=3.0, min_done=None):
    """rollout THROUGH the reference's episode handling: the environment's own termination rules are active and
    episode_length = L, stochastic_init off (restarts are deterministic), half of the environments start their episode three
    steps late; records done / progress / obs_before_reset per step and the gradient of a loss that reads obs,
    obs_before_reset and the rewards"""
    cls, mmf, _, _, has_et = CONFIGS[name]
    torch.manual_seed(0)
    np.random.seed(0)
    kw = dict(num_envs=n, device="cpu", render=False, seed=0, episode_length=L, no_grad=False, stochastic_init=False,
              MM_caching_frequency=mmf)
    if has_et:
        kw["early_termination"] = True
    env = getattr(envs, cls)(**kw)
    env.clear_grad()
    env.reset()
    env.progress_buf[: n // 2] = 3
    g = torch.Generator().manual_seed(11)
    q0, qd0 = env.get_state()
    prog0 = t2n(env.progress_buf)
    obs0 = env.initialize_trajectory()
    acts = torch.tanh(act_gain * (2.0 * torch.rand((H, n, env.num_actions), generator=g) - 1.0)).clone().requires_grad_(True)
    w = torch.randn((n, env.num_obs), generator=g)
    rec = dict(obs=[], rew=[], done=[], obs_before=[], progress=[])
    loss = 0.0
    for t in range(H):
        obs, rew, done, info = env.step(acts[t])
        loss = loss - rew.sum() + 0.01 * (w * info["obs_before_reset"]).sum() + 0.01 * (w * obs).sum()
        rec["obs"].append(t2n(obs)); rec["rew"].append(t2n(rew)); rec["done"].append(t2n(done))
        rec["obs_before"].append(t2n(info["obs_before_reset"])); rec["progress"].append(t2n(env.progress_buf))
    loss.backward()
    assert sum(int(d.sum()) for d in rec["done"]) >= (2 * n if min_done is None else min_done)
    return dict(q0=t2n(q0).reshape(n, -1), qd0=t2n(qd0).reshape(n, -1), progress0=prog0, obs0=t2n(obs0),
                actions=t2n(acts), w=t2n(w), grad_actions=t2n(acts.grad), loss=np.float64(loss.item()),
                episode_length=L, mm_freq=mmf, q_final=t2n(env.state.joint_q).reshape(n, -1),
                **{k: np.stack(v) for k, v in rec.items()})


def exploded_golden(envs, name="humanoid", n=4, H=6):
    """rollout in which environments blow up: env 1 starts with a finite but absurd root velocity (|qd| > 1e6: the
    'invalid value' rule of humanoid.py:340-356 fires at the first step), env 3 with one that overflows to inf / NaN inside
    the step.  Loss = -sum(rew) + 0.01 sum(w * obs_before_reset) with the observation of an invalid environment masked out, as
    algorithms/shac.py:205-213 does before it feeds obs_before_reset to the critic.  Records what the reference returns
    (rewards forced to 0, done flags) and the action gradients after its nan_to_num hooks (humanoid.py:195-206)."""
    cls, mmf, _, _, has_et = CONFIGS[name]
    torch.manual_seed(0)
    np.random.seed(0)
    env = getattr(envs, cls)(num_envs=n, device="cpu", render=False, seed=0, episode_length=1000, no_grad=False,
                             stochastic_init=False, MM_caching_frequency=mmf)
    env.clear_grad()
    env.reset()
    q0, qd0 = env.get_state()
    qd0 = qd0.view(n, -1).clone()
    qd0[1, 3] = 3.0e6
    qd0[3, 4] = 1.0e30
    env.reset_with_state(q0, qd0.view(-1))
    g = torch.Generator().manual_seed(5)
    obs0 = env.initialize_trajectory()
    acts = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=g) - 1.0).clone().requires_grad_(True)
    w = torch.randn((n, env.num_obs), generator=g)
    rec = dict(rew=[], done=[], progress=[], obs=[])
    loss = 0.0
    for t in range(H):
        obs, rew, done, info = env.step(acts[t])
        ob = info["obs_before_reset"]
        bad = (torch.isnan(ob).sum(-1) > 0) | (torch.isinf(ob).sum(-1) > 0) | ((ob.abs() > 1e6).sum(-1) > 0)
        loss = loss - rew.sum() + 0.01 * (w * torch.where(bad.unsqueeze(-1), torch.zeros_like(ob), ob)).sum()
        rec["rew"].append(t2n(rew)); rec["done"].append(t2n(done)); rec["progress"].append(t2n(env.progress_buf))
        rec["obs"].append(t2n(obs))
    loss.backward()
    return dict(q0=t2n(q0).reshape(n, -1), qd0=t2n(qd0), obs0=t2n(obs0), actions=t2n(acts), w=t2n(w),
                grad_actions=t2n(acts.grad), loss=np.float64(loss.item()), mm_freq=mmf,
                q_final=t2n(env.state.joint_q).reshape(n, -1), **{k: np.stack(v) for k, v in rec.items()})


def ant_extra_goldens(df, envs):
    """H = 32 rollout, and a rollout through the reference's own reset logic (see module docstring)"""
    out = {}
    # (a) BASELINE horizon
    n, H = 8, 32
    CONFIGS["ant"] = ("AntEnv", 16, n, H, True)
    out["ant_rollout_h32"] = rollout_golden(df, envs, "ant")
    CONFIGS["ant"] = ("AntEnv", 16, 4, 6, True)
    # (b) episode handling
    out["ant_episode"] = episode_golden(envs, "ant", n=8, H=24, L=10)
    return out


def main():
    names = sys.argv[1:] or ["cartpole", "ant", "humanoid", "snu"]
    df, envs = ref_harness.load_reference()
    os.makedirs(OUT, exist_ok=True)
    if "h32_extra" in names:
        # BASELINE.json's horizon for the two humanoid configurations (2 environments each: the reference CPU path does
        # ~15-40 env-steps/s on them)
        names.remove("h32_extra")
        for name in ("humanoid", "snu"):
            saved = CONFIGS[name]
            CONFIGS[name] = (saved[0], saved[1], 2, 32, saved[4])
            np.savez_compressed(os.path.join(OUT, name + "_rollout_h32.npz"), **rollout_golden(df, envs, name))
            CONFIGS[name] = saved
            print("golden written:", name + "_rollout_h32")
    if "cartpole_64x16" in names:
        # BASELINE.json configs[0] literally: CartPoleSwingUp, 64 environments, H = 16
        names.remove("cartpole_64x16")
        saved = CONFIGS["cartpole"]
        CONFIGS["cartpole"] = (saved[0], saved[1], 64, 16, saved[4])
        np.savez_compressed(os.path.join(OUT, "cartpole_rollout_64x16.npz"), **rollout_golden(df, envs, "cartpole"))
        CONFIGS["cartpole"] = saved
        print("golden written: cartpole_rollout_64x16")
    if "ant_1024x32" in names:
        # BASELINE.json configs[1] literally (Ant, 1024 environments, H = 32) through the reference: ~1 min of its CPU path.
        # Kept small: the actions are regenerated from their seed by the tests (same CPU generator), observations are dropped,
        # action gradients are kept for every 2nd environment (all environments are independent; round 5: every 8th before),
        # rewards for all.
        names.remove("ant_1024x32")
        saved = CONFIGS["ant"]
        CONFIGS["ant"] = (saved[0], saved[1], 1024, 32, saved[4])
        g = rollout_golden(df, envs, "ant")
        CONFIGS["ant"] = saved
        slim = dict(q0=g["q0"], qd0=g["qd0"], rew=g["rew"], q_final=g["q_final"], mm_freq=g["mm_freq"], loss=g["loss"],
                    grad_actions_strided=g["grad_actions"][:, ::2], stride=np.int64(2), actions_check=g["actions"][:, :4],
                    action_seed=np.int64(3), preroll=np.int64(20))
        np.savez_compressed(os.path.join(OUT, "ant_1024x32.npz"), **slim)
        print("golden written: ant_1024x32")
    for tag, name, n, stride in (("humanoid_1024x32", "humanoid", 1024, 2), ("snu_512x32", "snu", 512, 4)):
        if tag in names:
            # BASELINE.json configs[2] / configs[3] literally, through the reference with its termination rules active
            # (random actions make humanoids fall: restarts are part of the recording).  Slimmed like ant_1024x32: actions and
            # loss weights are regenerated from the seed by the tests, observations dropped, gradients for every `stride`-th env.
            names.remove(tag)
            g = episode_golden(envs, name, n=n, H=32, L=1000, act_gain=1.0, min_done=0)
            slim = dict(q0=g["q0"][:1], qd0=g["qd0"][:1], progress0=g["progress0"], rew=g["rew"], done=g["done"].astype(np.int8),
                        progress=g["progress"].astype(np.int16), q_final=g["q_final"], loss=g["loss"], mm_freq=g["mm_freq"],
                        episode_length=g["episode_length"], grad_actions_strided=g["grad_actions"][:, ::stride],
                        actions_check=g["actions"][:, :2], w_check=g["w"][:2], stride=np.int64(stride), act_gain=np.float64(1.0),
                        seed=np.int64(11))
            np.savez_compressed(os.path.join(OUT, tag + ".npz"), **slim)
            print("golden written:", tag, "restarts recorded:", int(g["done"].sum()))
    if "mm1" in names:
        # MM_caching_frequency = 1, the constructor default of every environment class (envs/ant.py:32): the mass matrix is
        # rebuilt and its adjoint runs in EVERY substep (sim.py:2113, 2475)
        names.remove("mm1")
        for name, n, H in (("ant", 8, 8), ("snu", 2, 4), ("humanoid", 2, 4)):
            saved = CONFIGS[name]
            CONFIGS[name] = (saved[0], 1, n, H, saved[4])
            np.savez_compressed(os.path.join(OUT, name + "_rollout_mm1.npz"), **rollout_golden(df, envs, name))
            CONFIGS[name] = saved
            print("golden written:", name + "_rollout_mm1")
    if "exploded" in names:
        names.remove("exploded")
        np.savez_compressed(os.path.join(OUT, "humanoid_exploded.npz"), **exploded_golden(envs))
        print("golden written: humanoid_exploded")
    if "episodes_extra" in names:
        # the other environments' termination rules through the reference (humanoid: height + invalid-state checks,
        # humanoid.py:340-356; hopper: height, hopper.py:288-293; cartpole / cheetah: episode length only)
        names.remove("episodes_extra")
        for name, n, H, L in (("humanoid", 4, 14, 6), ("snu", 4, 14, 6), ("hopper", 8, 24, 10), ("cheetah", 8, 24, 10),
                              ("cartpole", 8, 24, 10)):
            np.savez_compressed(os.path.join(OUT, name + "_episode.npz"), **episode_golden(envs, name, n, H, L))
            print("golden written:", name + "_episode")
    if "ant_extra" in names:
        names.remove("ant_extra")
        for k, v in ant_extra_goldens(df, envs).items():
            np.savez_compressed(os.path.join(OUT, k + ".npz"), **v)
            print("golden written:", k)
    for name in names:
        cls, mmf, n, H, _ = CONFIGS[name]
        env = make_env(envs, name, 2, no_grad=True, stochastic=False)
        np.savez_compressed(os.path.join(OUT, name + "_model.npz"), **dump_model(env, 2))
        np.savez_compressed(os.path.join(OUT, name + "_step.npz"), **step_golden(df, envs, name))
        np.savez_compressed(os.path.join(OUT, name + "_rollout.npz"), **rollout_golden(df, envs, name))
        print("golden written:", name)


if __name__ == "__main__":
    main()
