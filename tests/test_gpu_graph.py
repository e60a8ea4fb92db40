"""GPU: whole-trajectory HIP-graph submission (diffrl_amd/graph.py): the captured rollout replays to
the same losses and gradients as the eager DFlexEnv.step loop."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ant(n, **kw):
    from diffrl_amd import envs
    args = dict(num_envs=n, device="cuda:0", no_grad=False, stochastic_init=False, MM_caching_frequency=16,
                early_termination=True, episode_length=1000)
    args.update(kw)
    return envs.AntEnv(**args)


def test_open_loop_replay_is_bit_identical_to_eager():
    from diffrl_amd.graph import GraphedRollout
    dev, n, H = torch.device("cuda:0"), 64, 8
    gen = torch.Generator().manual_seed(0)
    actions = torch.tanh(2.0 * torch.rand((H, n, 8), generator=gen) - 1.0).to(dev)

    def body_for(a):
        def body(env):
            env.initialize_trajectory()
            rews = [env.step(a_t)[1] for a_t in a.unbind(0)]
            return -torch.stack(rews).sum()
        return body

    e1 = _ant(n)
    e1.reset()
    a1 = actions.clone().requires_grad_(True)
    loss1 = body_for(a1)(e1)
    loss1.backward()
    e2 = _ant(n)
    e2.reset()
    a2 = actions.clone().requires_grad_(True)
    roll = GraphedRollout(e2, body_for(a2), leaves=[a2], carry_state=False)
    for _ in range(2):
        loss2 = roll.replay()
    torch.cuda.synchronize()
    assert float(loss2) == float(loss1.detach())
    assert torch.equal(a2.grad, a1.grad)


def test_policy_in_the_loop_with_restarts_and_carried_state():
    """actor MLP inside the captured loop, episodes ending (and restarting in-kernel) inside the graph, two consecutive
    rollouts that continue the trajectory: replays == eager"""
    from diffrl_amd.graph import GraphedRollout
    dev, n, H = torch.device("cuda:0"), 32, 6
    torch.manual_seed(0)
    actor0 = torch.nn.Sequential(torch.nn.Linear(37, 32), torch.nn.ELU(), torch.nn.Linear(32, 8)).to(dev)

    def make(actor):
        def body(env):
            obs = env.initialize_trajectory()
            total = 0.0
            for t in range(H):
                obs, rew, done, info = env.step(torch.tanh(actor(obs)))
                # a graph-friendly use of `done`: a mask, not an index list
                total = total - (rew * (1.0 - 0.5 * done.float())).sum() + 1e-3 * info["obs_before_reset"].pow(2).sum()
            return total
        return body

    import copy
    res = []
    for graphed in (False, True):
        actor = copy.deepcopy(actor0)
        e = _ant(n, episode_length=4)      # every environment finishes (length) inside each rollout
        e.reset()
        body = make(actor)
        out = []
        if graphed:
            roll = GraphedRollout(e, body, leaves=list(actor.parameters()), carry_state=True)
        for it in range(3):
            if graphed:
                loss = roll.replay()
            else:
                for p in actor.parameters():
                    p.grad = None
                loss = body(e)
                loss.backward()
            torch.cuda.synchronize()
            out.append((float(loss), [p.grad.clone() for p in actor.parameters()]))
        if graphed:
            roll.sync_env()
        res.append((out, e.state.joint_q.detach().clone(), e.progress_buf.clone()))
    for (l1, g1), (l2, g2) in zip(res[0][0], res[1][0]):
        assert abs(l1 - l2) <= 1e-5 * abs(l1)
        for a, b in zip(g1, g2):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)
    assert res[0][0][0][0] != res[0][0][1][0]        # the second rollout really continued from a different state
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6) and torch.equal(res[0][2], res[1][2])


def test_stochastic_restarts_inside_a_graph():
    """the start-state pool is redrawn inside the captured region (clear_grad in the body): replays draw new states"""
    from diffrl_amd.graph import GraphedRollout
    dev, n = torch.device("cuda:0"), 32
    e = _ant(n, stochastic_init=True, episode_length=2, no_grad=False)
    e.reset()
    a = torch.zeros((2, n, 8), device=dev, requires_grad=True)

    def body(env):
        env.initialize_trajectory()
        rews = [env.step(a_t)[1] for a_t in a.unbind(0)]
        return -torch.stack(rews).sum()

    roll = GraphedRollout(e, body, leaves=[a], carry_state=True)
    roll.replay()
    roll.sync_env()
    q1 = e.state.joint_q.detach().clone()
    roll.replay()
    roll.sync_env()
    q2 = e.state.joint_q.detach().clone()
    torch.cuda.synchronize()
    assert int(e.progress_buf.sum()) == 0          # every environment was restarted at the end of the rollout
    assert not torch.equal(q1, q2)                 # ... from a freshly drawn start state
    assert torch.isfinite(a.grad).all()


def test_policy_learns_through_graph_replays():
    """examples/shac_lite.py: 40 Adam steps on the actor through H=32 graph-replayed rollouts (stochastic restarts, policy in
    the loop) raise the mean reward -- the gradients that come out of the replays are the useful ones"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import shac_lite
    hist = shac_lite.main(["--iters", "40", "--envs", "128", "--graph", "--seed", "1"])
    assert sum(hist[-6:]) / 6 > sum(hist[:6]) / 6 + 0.15


def test_capture_after_an_eager_rollout_with_the_same_parameters():
    """regression: an eager rollout + backward on the default stream leaves autograd nodes alive through the environment's
    observation / reward buffers; GraphedRollout has to drop them before capturing (it used to segfault in capture_end)"""
    from diffrl_amd.graph import GraphedRollout
    dev, n = torch.device("cuda:0"), 64
    torch.manual_seed(0)
    e = _ant(n, stochastic_init=True)
    actor = torch.nn.Sequential(torch.nn.Linear(37, 32), torch.nn.ELU(), torch.nn.Linear(32, 8)).to(dev)

    def body(env):
        obs, loss = env.initialize_trajectory(), 0.0
        for _ in range(8):
            obs, rew, done, info = env.step(torch.tanh(actor(obs)))
            loss = loss - rew.sum()
        return loss / (8 * n)

    e.reset()
    body(e).backward()                     # eager, default stream
    g_eager = [p.grad.clone() for p in actor.parameters()]
    roll = GraphedRollout(e, body, leaves=list(actor.parameters()))
    roll.replay()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in actor.parameters())
    assert all(p.grad.shape == g.shape for p, g in zip(actor.parameters(), g_eager))
