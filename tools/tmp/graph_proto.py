import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
n, H = 1024, 32
env = bench.make_env("ant", n, "cuda:0")
gen = torch.Generator().manual_seed(1)
actions = torch.tanh(2.0 * torch.rand((H, n, env.num_actions), generator=gen) - 1.0).to(dev)
# eager reference
g_ref = bench.rollout(env, actions).clone()
torch.cuda.synchronize()
# static inputs
acts = actions.clone().requires_grad_(True)
env.reset(); env.clear_grad()
q0, qd0 = env.state.joint_q.clone(), env.state.joint_qd.clone()
prog0 = env.progress_buf.clone()

def body():
    env.state.joint_q = q0.clone(); env.state.joint_qd = qd0.clone()
    env.progress_buf.copy_(prog0)
    rews = []
    for a_t in acts.unbind(0):
        obs, rew, done, info = env.step(a_t)
        rews.append(rew)
    loss = -torch.stack(rews).sum()
    loss.backward()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        acts.grad = None
        body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager-in-side-stream grad err", float((acts.grad - g_ref).abs().max()))
g = torch.cuda.CUDAGraph()
acts.grad = None
with torch.cuda.graph(g):
    loss = body()
torch.cuda.synchronize()
print("captured")
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("replay grad err", float((acts.grad - g_ref).abs().max()), "loss", float(loss))
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("graph replay: %.3f ms/rollout -> %.3e env-steps/s" % (dt * 1e3, n * H / dt))
