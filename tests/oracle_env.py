"""Test-only: runs diffrl_amd's (unfused, torch) environment surface on the CPU with the scalar ORACLE as
integrator, so that the env protocol / observation / reward code can be checked against the reference
rollouts without a GPU, and so that gradient conditioning can be probed in the reference's own operation
order.  (The product never does this: without the HIP library `env.step` raises.)"""
import numpy as np
import torch

from diffrl_amd import envs
from diffrl_amd.dflex.model import State
from oracle_lib import oracle_backward, oracle_forward

SUBSTEPS = {"cartpole": 4, "ant": 16, "humanoid": 48, "snu": 48, "hopper": 16, "cheetah": 16}
MM = {"cartpole": 4, "ant": 16, "humanoid": 48, "snu": 8, "hopper": 16, "cheetah": 16}
CLS = {"cartpole": envs.CartPoleSwingUpEnv, "ant": envs.AntEnv, "humanoid": envs.HumanoidEnv,
       "snu": envs.SNUHumanoidEnv, "hopper": envs.HopperEnv, "cheetah": envs.CheetahEnv}


def make_cpu_env(name, n, template, episode_length=1000, early_termination=False):
    kw = dict(num_envs=n, device="cpu", render=False, seed=0, episode_length=episode_length, no_grad=False,
              stochastic_init=False, MM_caching_frequency=MM[name])
    if name in ("cartpole", "ant", "hopper", "cheetah"):
        kw["early_termination"] = early_termination
    e = CLS[name](**kw)
    e.fused = False
    t, S, mm, dt = template, SUBSTEPS[name], MM[name], 1.0 / 60.0

    class OracleStep(torch.autograd.Function):
        @staticmethod
        def forward(ctx, q, qd, act, mact):
            ctx.inp = (q.detach().numpy().reshape(n, -1), qd.detach().numpy().reshape(n, -1),
                       act.detach().numpy().reshape(n, -1),
                       mact.detach().numpy().reshape(n, -1) if mact is not None else None)
            qo, qdo, _ = oracle_forward(t, *ctx.inp, dt, S, mm)
            return torch.tensor(qo).reshape(-1), torch.tensor(qdo).reshape(-1)

        @staticmethod
        def backward(ctx, gq, gqd):
            r = oracle_backward(t, *ctx.inp, dt, S, mm, gq.numpy().reshape(n, -1), gqd.numpy().reshape(n, -1))
            gm = torch.tensor(r["gmact"]).reshape(-1) if ctx.inp[3] is not None else None
            return (torch.tensor(r["gq"]).reshape(-1), torch.tensor(r["gqd"]).reshape(-1),
                    torch.tensor(r["gact"]).reshape(-1), gm)

    class OracleIntegrator:
        def forward(self, model, state, dt_, substeps, mm_freq):
            assert substeps == S and mm_freq == mm
            st = State(act_like=model.joint_qd)
            mact = model.muscle_activation if model.muscle_count else None
            st.joint_q, st.joint_qd = OracleStep.apply(state.joint_q, state.joint_qd, state.joint_act, mact)
            return st

    e.integrator = OracleIntegrator()
    return e


def rollout_grad(name, template, q0, qd0, actions):
    """obs, rew per step and d(-sum rew)/d actions through the torch env surface + oracle"""
    H, n = actions.shape[0], actions.shape[1]
    e = make_cpu_env(name, n, template)
    e.clear_grad()
    e.reset()
    e.reset_with_state(torch.tensor(q0, dtype=torch.float32).reshape(-1), torch.tensor(qd0, dtype=torch.float32).reshape(-1))
    e.initialize_trajectory()
    acts = torch.tensor(actions, requires_grad=True)
    loss, obs_l, rew_l = 0.0, [], []
    for s in range(H):
        obs, rew, done, info = e.step(acts[s])
        assert int(done.sum()) == 0
        obs_l.append(obs.detach().numpy().copy())
        rew_l.append(rew.detach().numpy().copy())
        loss = loss - rew.sum()
    loss.backward()
    return np.stack(obs_l), np.stack(rew_l), acts.grad.numpy().copy()


def episode_rollout_grad(name, template, progress0, actions, w, episode_length, q0_scale=None):
    """The loss of the <env>_<N>x32 recordings (oracle/gen_golden.py: episode_golden) for a sub-batch of environments, through
    the torch env surface WITH its termination rules and restarts + the oracle: d loss / d actions, done flags.
    q0_scale: relative perturbation of the start state (1-ulp conditioning probe in the reference's operation order)."""
    H, n = actions.shape[0], actions.shape[1]
    e = make_cpu_env(name, n, template, episode_length=episode_length, early_termination=True)
    e.clear_grad()
    e.reset()
    if q0_scale is not None:
        q0, qd0 = e.get_state()
        e.reset_with_state((q0.view(n, -1) * torch.tensor(q0_scale, dtype=torch.float32)).reshape(-1), qd0)
    e.progress_buf[:] = torch.tensor(progress0)
    e.initialize_trajectory()
    acts = torch.tensor(actions, requires_grad=True)
    wt = torch.tensor(w)
    loss, dones = 0.0, []
    for s in range(H):
        obs, rew, done, info = e.step(acts[s])
        loss = loss - rew.sum() + 0.01 * (wt * info["obs_before_reset"]).sum() + 0.01 * (wt * obs).sum()
        dones.append(done.numpy().copy())
    loss.backward()
    return acts.grad.numpy().copy(), np.stack(dones)
