"""Whole-trajectory submission: H x (policy -> env.step) plus the backward sweep captured
once as a HIP graph and replayed as ONE submission per rollout.

The reference drives the same loop from Python, one kernel launch at a time (algorithms/shac.py:184-251: actor MLP,
env.step, reward bookkeeping per step, then loss.backward()).  With the simulation fused into one launch per env.step and
the episode bookkeeping done in-kernel, nothing in `DFlexEnv.step` synchronises with the host any more, so the whole
rollout -- including whatever torch modules the caller puts in the loop -- is capturable: the launches made through the
C ABI go to torch's current stream, which is the capturing stream during `torch.cuda.graph`.

    roll = GraphedRollout(env, body)      # body(env) -> scalar loss; runs H x env.step, policy in the loop
    loss = roll.replay()                  # forward + backward, one hipGraphLaunch; .grad of the leaves is refreshed

`body` must be capturable: no `.item()`, no `nonzero()`, no Python branching on device data (`done` has to be used as a
mask, not as an index list).  If the same leaves took part in an EAGER backward before, drop every tensor of that rollout
(losses, observations, rewards) first: live autograd nodes bound to the default stream break the capture.  The leaves whose `.grad` the caller reads (actor parameters, an action tensor) keep their
identity; their `.grad` tensors live in the graph's memory pool and are overwritten by every replay.
"""
import torch


class GraphedRollout:
    def __init__(self, env, body, leaves=(), carry_state=True, warmup=3, backward=True):
        """env: a fused DFlexEnv on a GPU; body(env) -> loss (0-dim tensor).
        leaves: the tensors / parameters whose .grad the caller reads; their .grad is reset to None before the capture
        so that every replay ASSIGNS fresh gradients (in the graph's memory pool) instead of accumulating.
        carry_state: a replay starts where the previous one ended (as consecutive SHAC rollouts do); False: every replay
        restarts from the state the environment had at construction (deterministic benchmark).
        backward: also capture loss.backward()."""
        if torch.device(env.device).type != "cuda":
            raise RuntimeError("GraphedRollout needs a GPU environment")
        self.env, self.body, self.carry, self.backward = env, body, carry_state, backward
        # Autograd nodes of an earlier eager rollout that are still alive (kept by the environment's observation / reward
        # buffers, or by a loss the caller still holds) carry AccumulateGrad nodes bound to the DEFAULT stream; re-using
        # those inside a capture is fatal on this stack (segfault in hipStreamEndCapture).  Drop the environment's
        # references; the caller must not keep losses / observations of earlier rollouts either.
        import gc
        env.detach_buffers()
        gc.collect()
        with torch.no_grad():
            self._q = env.state.joint_q.detach().clone()
            self._qd = env.state.joint_qd.detach().clone()
            self._act = env.actions.detach().clone()
            self._progress = env.progress_buf.clone()
        self._frames_per_replay = None
        self._count0 = None
        # Finished environments restart inside the captured step kernels: the start state is perturbed with counter-based
        # random numbers keyed by the per-environment restart counter, which lives in device memory and advances with every
        # replay -- nothing has to be drawn on the host (the reference's reset_state() indexed writes are not capturable).
        # (an environment captured on its unfused torch path has no in-kernel restarts and no pool: nothing to build or pin)
        if getattr(env, "fused", False):
            env._episode_io()
        if not carry_state and getattr(env, "_reset_count", None) is not None:
            # the in-kernel restart noise is keyed by the per-environment restart counter, which every replay advances: a
            # replay that restarts from the construction state must restart the counters too, or replays with
            # stochastic_init would draw different start states ("deterministic benchmark")
            self._count0 = env._reset_count.clone()
        # warm-up on a side stream (lazy initialisation, allocator pools), as torch.cuda.graph requires
        side = torch.cuda.Stream(device=env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._run_once(write_back=False)
        torch.cuda.current_stream(env.device).wait_stream(side)
        torch.cuda.synchronize(env.device)
        # The last warm-up rollout's autograd graph is still alive through the environment's observation / reward buffers, and
        # with it the AccumulateGrad nodes of the leaves, bound to the stream they were created on.  Drop it, and capture on the
        # SAME side stream the warm-up ran on: a leaf's accumulator that survives anyway (the caller holds a loss) then still
        # matches the stream of the captured backward (no "AccumulateGrad node's stream does not match" synchronisation
        # inside the replays).
        env.detach_buffers()
        gc.collect()
        self.leaves = list(leaves)
        for t in self.leaves:
            t.grad = None
        f0 = env.num_frames
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: other threads (e.g. the RCCL watchdog of a torch.distributed job) may keep making HIP calls
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.loss = self._run_once(write_back=carry_state)
        self._frames_per_replay = env.num_frames - f0
        torch.cuda.synchronize(env.device)
        # the captured launches hold the pointers of the environment's start-state pool and the noise settings: from now on a
        # change of stochastic_init / start_* raises in DFlexEnv._episode_io instead of being silently ignored by replays
        env._pool_pinned = True

    def _run_once(self, write_back):
        env = self.env
        # start from the static copies (fresh tensors: the body builds a new autograd graph on them)
        st = type(env.state)(act_like=env.model.joint_qd, model=env.model)
        st.joint_q, st.joint_qd = self._q.clone(), self._qd.clone()
        env.state = st
        env.actions = self._act.clone()
        env.progress_buf = self._progress.clone()
        if self._count0 is not None:
            env._reset_count.copy_(self._count0)
        loss = self.body(env)
        if self.backward:
            loss.backward()
        if write_back:
            with torch.no_grad():
                self._q.copy_(env.state.joint_q.detach())
                self._qd.copy_(env.state.joint_qd.detach())
                self._act.copy_(env.actions.detach())
                self._progress.copy_(env.progress_buf)
        return loss.detach()

    def replay(self):
        """re-executes the captured rollout (forward + backward); returns the static loss tensor"""
        self.env._check_pinned_pool()   # raises if the start-state settings changed since the capture (a key comparison only)
        self.graph.replay()
        if self._frames_per_replay:
            self.env.num_frames += self._frames_per_replay
            self.env.sim_time += self._frames_per_replay * self.env.sim_dt
        return self.loss

    def sync_env(self):
        """after replays with carry_state: point the environment object at the carried state (for eager use afterwards)"""
        env = self.env
        with torch.no_grad():
            st = type(env.state)(act_like=env.model.joint_qd, model=env.model)
            st.joint_q, st.joint_qd = self._q.clone(), self._qd.clone()
            env.state = st
            env.actions = self._act.clone()
            env.progress_buf = self._progress.clone()
