"""Developer probe (GPU box): the headline parity case (tests/test_reference_episodes.py::test_gpu_baseline_config_vs_reference)
with the library in $DSIM_LIB -- per-environment gradient error against the reference recording."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle_lib import golden, relerr
from test_reference_episodes import _ant_1024x32_actions
from diffrl_amd import envs
g = golden("ant_1024x32")
n, H = g["q0"].shape[0], 32
dev = torch.device("cuda:0")
e = envs.AntEnv(num_envs=n, device="cuda:0", no_grad=False, stochastic_init=False, MM_caching_frequency=16, early_termination=False, episode_length=1000)
e.reset()
e.reset_with_state(torch.tensor(g["q0"], device=dev).reshape(-1), torch.tensor(g["qd0"], device=dev).reshape(-1))
e.initialize_trajectory()
acts = _ant_1024x32_actions(g).to(dev).requires_grad_(True)
rews = []
for t in range(H):
    obs, rew, done, info = e.step(acts[t]); rews.append(rew)
loss = -torch.stack(rews).sum(); loss.backward()
stride = int(g["stride"])
a = acts.grad[:, ::stride].cpu().numpy().astype(np.float64); r = g["grad_actions_strided"].astype(np.float64)
per = np.abs(a - r).max(axis=(0, 2)) / (np.abs(r).max(axis=(0, 2)) + 1e-30)   # per environment, relative to its own largest gradient
order = np.argsort(-per)[:6]
print(os.environ.get("DSIM_LIB", "product"), "relerr %.3e" % relerr(a, r), "worst envs (index/stride, err):", [(int(i), "%.2e" % per[i]) for i in order],
      "n>1e-3: %d of %d" % (int((per > 1e-3).sum()), len(per)), "n>5e-3: %d" % int((per > 5e-3).sum()), "n>2e-2: %d" % int((per > 2e-2).sum()), "rew err %.2e" % (np.abs(torch.stack(rews).detach().cpu().numpy() - g["rew"]).max()))
