"""Developer tool: where does the wall time of a rollout go (host launch path vs GPU kernels)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device("cuda:0")
env = bench.make_env("ant", 1024, "cuda:0")
H = 32
gen = torch.Generator().manual_seed(1)
actions = torch.tanh(2.0 * torch.rand((H, 1024, env.num_actions), generator=gen) - 1.0).to(dev)
for _ in range(3):
    bench.rollout(env, actions)
torch.cuda.synchronize()
for rep in range(3):
    torch.cuda.synchronize(); ta = time.perf_counter()
    env.clear_grad(); torch.cuda.synchronize(); tb = time.perf_counter()
    env.reset(); torch.cuda.synchronize(); tc = time.perf_counter()
    env.initialize_trajectory(); torch.cuda.synchronize(); td = time.perf_counter()
    print("clear_grad %.2f ms  reset %.2f ms  initialize_trajectory %.2f ms" % ((tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3))
    acts = actions.detach().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = 0.0
    for t in range(H):
        obs, rew, done, info = env.step(acts[t])
        loss = loss - rew.sum()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print("fwd: host %.2f ms, +drain %.2f ms | bwd: host %.2f ms, +drain %.2f ms | total %.2f ms" %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3))
import cProfile, pstats
env.clear_grad(); env.reset(); env.initialize_trajectory()
acts = actions.detach().requires_grad_(True)
pr = cProfile.Profile(); pr.enable()
loss = 0.0
for t in range(H):
    obs, rew, done, info = env.step(acts[t]); loss = loss - rew.sum()
loss.backward(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
