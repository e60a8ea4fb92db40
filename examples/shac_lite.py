"""Short-horizon policy optimisation through the differentiable simulator (the actor half of SHAC, algorithms/shac.py:
169-300, without the critic): H-step rollouts with the policy in the loop, truncated back-propagation through time,
Adam.  With --graph the whole rollout (policy, env.step, loss, backward) is one HIP-graph submission per iteration.

    python examples/shac_lite.py --env ant --envs 256 --iters 60 --graph
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="ant")
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args(argv)

    from diffrl_amd import envs
    from diffrl_amd.graph import GraphedRollout
    cls = {"ant": envs.AntEnv, "hopper": envs.HopperEnv, "cheetah": envs.CheetahEnv, "humanoid": envs.HumanoidEnv}[a.env]
    mm = {"ant": 16, "hopper": 16, "cheetah": 16, "humanoid": 48}[a.env]
    torch.manual_seed(a.seed)
    env = cls(num_envs=a.envs, device="cuda:0", no_grad=False, stochastic_init=True, MM_caching_frequency=mm,
              episode_length=1000, seed=a.seed)
    dev = torch.device("cuda:0")
    actor = torch.nn.Sequential(torch.nn.Linear(env.num_obs, 128), torch.nn.ELU(), torch.nn.Linear(128, 64), torch.nn.ELU(),
                                torch.nn.Linear(64, env.num_actions)).to(dev)
    opt = torch.optim.Adam(actor.parameters(), lr=a.lr, betas=(0.7, 0.95), capturable=a.graph)
    H, n = a.horizon, a.envs
    stat = torch.zeros(1, device=dev)     # mean reward per step of the last rollout

    def body(e):
        obs = e.initialize_trajectory()
        disc = torch.ones(n, device=dev)
        total, rsum = 0.0, 0.0
        for t in range(H):
            obs, rew, done, info = e.step(torch.tanh(actor(obs)))
            total = total - (disc * rew).sum()
            rsum = rsum + rew.detach().mean()
            disc = torch.where(done.bool(), torch.ones_like(disc), disc * a.gamma)   # restart the discount with the episode
        stat.copy_((rsum / H).reshape(1))
        return total / (n * H)

    env.reset()
    roll = GraphedRollout(env, body, leaves=list(actor.parameters()), carry_state=True) if a.graph else None
    hist = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(a.iters):
        if roll is not None:
            roll.replay()
        else:
            opt.zero_grad(set_to_none=True)
            body(env).backward()
        torch.nn.utils.clip_grad_norm_(actor.parameters(), 1.0)
        opt.step()
        hist.append(stat.clone())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    hist = torch.cat(hist).cpu().tolist()
    k = max(1, a.iters // 6)
    print("mean reward/step: first %d iters %.3f, last %d iters %.3f; %.1f ms per iteration (%s), %.2e env-steps/s"
          % (k, sum(hist[:k]) / k, k, sum(hist[-k:]) / k, el / a.iters * 1e3, "graph" if a.graph else "eager",
             a.iters * n * H / el))
    return hist


if __name__ == "__main__":
    main()
