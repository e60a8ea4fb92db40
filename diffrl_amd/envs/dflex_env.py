"""DFlexEnv surface (envs/dflex_env.py + the per-env step/reset/clear_grad protocol of the reference,
e.g. envs/ant.py:156-264) on top of the fused HIP step.

What `algorithms/shac.py` relies on and is kept verbatim: constructor keywords, `num_envs`, `num_obs`,
`num_actions`, `episode_length`, `step(actions) -> (obs, rew, done, extras)` with
`extras['obs_before_reset']` / `extras['episode_end']` when gradients are on, `reset`, `clear_grad`,
`initialize_trajectory`, `get_state` / `reset_with_state`, `get_checkpoint`; `obs` and `rew` carry
`grad_fn` back to the actions and to the previous state.

Structure differs from the reference (which repeats the protocol in every environment file): the
protocol lives here once; an environment supplies its asset, action mapping, observation and reward.
"""
import os

import numpy as np
import torch

from .. import dflex as df
from ..engine import EnvStep

ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def find_asset(name):
    """Original asset file if available ($DIFFRL_ASSETS or the package's assets/ dir), else None -> the
    environment falls back to its compiled asset (.npz builder snapshot).  No other location is searched: what a
    library call loads must not depend on what else happens to be on the box."""
    roots = [os.environ.get("DIFFRL_ASSETS"), ASSET_DIR]
    for r in roots:
        if r and os.path.exists(os.path.join(r, name)):
            return os.path.join(r, name)
    return None


class Box:
    """Minimal stand-in for gym.spaces.Box with gym's constructor (rl_games reads low / high / shape / dtype and builds
    Box(low=0, high=1, shape=(k,)) spaces of its own, rl_games/common/experience.py:335)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)   # (a numpy dtype object, as in gym: rl_games keys a dict with it)
        if shape is not None:
            low, high = np.full(shape, low, self.dtype), np.full(shape, high, self.dtype)
        self.low, self.high = np.asarray(low, self.dtype), np.asarray(high, self.dtype)
        self.shape = self.low.shape


class DFlexEnv:
    # subclasses set these
    sim_substeps = 16
    sanitize_grads = False   # nan_to_num hooks on the state / action gradients (humanoid.py:195-206)
    keep_act_on_clear = False
    fused = os.environ.get("DIFFRL_AMD_UNFUSED", "0") != "1"   # one launch per env.step

    def __init__(self, num_envs, num_obs, num_act, episode_length, MM_caching_frequency=1, seed=0, no_grad=True,
                 render=False, device="cuda:0"):
        self.seed = seed
        self.no_grad = no_grad
        df.config.no_grad = self.no_grad
        self.episode_length = episode_length
        self.device = device
        self.visualize = False
        if render:
            print("[diffrl_amd] USD rendering is not part of the MI355X hot path; render=True is ignored")
        self.sim_time = 0.0
        self.num_frames = 0
        self.num_environments = num_envs
        self.num_agents = 1
        self.MM_caching_frequency = MM_caching_frequency
        self.num_observations = num_obs
        self.num_actions = num_act
        self.obs_space = Box(np.ones(num_obs) * -np.inf, np.ones(num_obs) * np.inf)
        self.act_space = Box(np.ones(num_act) * -1.0, np.ones(num_act) * 1.0)
        dev = self.device
        self.obs_buf = torch.zeros((num_envs, num_obs), device=dev, dtype=torch.float)
        self.rew_buf = torch.zeros(num_envs, device=dev, dtype=torch.float)
        self.reset_buf = torch.ones(num_envs, device=dev, dtype=torch.long)
        self.termination_buf = torch.zeros(num_envs, device=dev, dtype=torch.long)
        self.progress_buf = torch.zeros(num_envs, device=dev, dtype=torch.long)
        self.actions = torch.zeros((num_envs, num_act), device=dev, dtype=torch.float)
        self.extras = {}
        self.dt = 1.0 / 60.0
        self.sim_dt = self.dt

    # ---- bookkeeping the callers read ------------------------------------------------------------
    def get_number_of_agents(self):
        return self.num_agents

    observation_space = property(lambda self: self.obs_space)
    action_space = property(lambda self: self.act_space)
    num_envs = property(lambda self: self.num_environments)
    num_acts = property(lambda self: self.num_actions)
    num_obs = property(lambda self: self.num_observations)

    # ---- model construction ----------------------------------------------------------------------
    def _finalize(self, builder, ground, gravity=(0.0, -9.81, 0.0)):
        """builder holds ONE articulation; the model replicates it num_envs times on the device."""
        builder.replicate(self.num_environments)
        self.builder = builder
        self.model = builder.finalize(self.device)
        self.model.ground = ground
        self.model.gravity = torch.tensor(gravity, dtype=torch.float32, device=self.device)
        self.integrator = df.sim.SemiImplicitIntegrator()
        self.state = self.model.state()
        if ground:
            self.model.collide(self.state)
        self.num_joint_q = self.model.coords_per_articulation
        self.num_joint_qd = self.model.dofs_per_articulation

    def _q(self):
        return self.state.joint_q.view(self.num_envs, -1)

    def _qd(self):
        return self.state.joint_qd.view(self.num_envs, -1)

    # ---- fused path: action mapping + observation + reward inside the step kernels ---------------
    def fused_spec(self):
        """capi.EnvSpec describing this environment's action mapping / observation / reward, or None."""
        return None

    def _spec(self):
        if getattr(self, "_spec_cache", None) is None:
            self._spec_cache = self.fused_spec()
        if self._spec_cache is not None:
            # the reference's nan_to_num hooks on joint_q / joint_qd / actions (humanoid.py:195-206): done by the adjoint
            # launch's own output stores (include/dsim.h: sanitize_grads), not by three torch kernels per step.  Refreshed on
            # every call (a host int store): the attribute may be toggled between steps, as on the unfused path -- the value a
            # step's BACKWARD sees is the one in force when that backward runs (a captured graph keeps the one it captured)
            self._spec_cache.sanitize_grads = int(bool(self.sanitize_grads))
        return self._spec_cache

    def stored_actions(self, actions):
        """what the environment keeps as self.actions for raw policy actions (clipped, maybe remapped)"""
        return torch.clip(actions, -1.0, 1.0)

    def _may_reset(self):
        """(unfused path) False when no environment can possibly be flagged this step: skips the device->host sync"""
        return True

    # termination rules the fused step applies in-kernel (the tail of the reference's calculateReward)
    height_terminate = False     # obs[:, 0] < termination_height
    check_invalid = False        # non-finite / exploded state -> reset with reward 0 (humanoid.py:340-356)
    def reset_noise(self):
        """(noise_q [n_q], noise_qd [n_qd], noise_angle): amplitudes of the uniform noise a stochastic restart adds to the
        start state, coordinate k gets + noise[k] * (u - 0.5), u ~ U[0, 1) -- the environment's reset_state() written
        as data, so that the fused step can draw a FRESH start state in the kernel for every restart (envs/ant.py:199-234
        does it with torch.rand per reset).  None: the environment has no stochastic reset."""
        return None

    def _deterministic_start_state(self):
        """[1][num_envs][nq], [1][num_envs][nd]: reset_state() of every environment without the stochastic part"""
        saved_state, saved_actions = self.state, self.actions
        saved_flag = getattr(self, "stochastic_init", False)
        ids = torch.arange(self.num_envs, dtype=torch.long, device=self.device)
        with torch.no_grad():
            st = self.model.state()
            st.joint_q, st.joint_qd = self.model.joint_q.clone(), self.model.joint_qd.clone()
            self.state = st
            self.stochastic_init = False
            try:
                self.reset_state(ids)
            finally:
                self.stochastic_init = saved_flag
            q, qd = self.state.joint_q.view(1, self.num_envs, -1).clone(), self.state.joint_qd.view(1, self.num_envs, -1).clone()
        self.state, self.actions = saved_state, saved_actions
        return q.contiguous(), qd.contiguous()

    def _start_state_key(self):
        """what the cached start states / noise amplitudes are built from (host-side only: identity + in-place version counter
        of the tensors, no device work, no host sync; the key's tensors are kept referenced, an id() alone could be recycled)"""
        def stamp(v):
            return (id(v), v._version) if torch.is_tensor(v) else v
        attrs = tuple(getattr(self, a, None) for a in ("start_pos", "start_rotation", "start_joint_q", "start_joint_target", "start_height"))
        self._pool_key_refs = attrs
        return (bool(getattr(self, "stochastic_init", False)),) + tuple(stamp(v) for v in attrs)

    def _check_pinned_pool(self, key=None):
        """raises if a GraphedRollout was captured on this environment and the start-state settings changed since: the captured
        launches hold the device pointers of the pool / noise tensors AND the launch arguments derived from them (noise on or
        off, the rotation amplitude), a replay cannot follow such a change.  A key comparison, nothing else: no EpisodeIO is
        built, no start state drawn (GraphedRollout.replay calls this on every replay)."""
        if not getattr(self, "_pool_pinned", False) or getattr(self, "_pool", None) is None:
            return
        if (key if key is not None else self._start_state_key()) != getattr(self, "_pool_key", None):
            raise RuntimeError("%s: stochastic_init or a start-state attribute changed after a GraphedRollout was captured on "
                               "this environment; build a new GraphedRollout (the captured launches keep the old start "
                               "states)" % type(self).__name__)

    def _episode_io(self):
        """EpisodeIO of the fused step: progress_buf, the deterministic start state of every environment and -- with
        stochastic resets -- the noise amplitudes the kernel perturbs it with (a fresh counter-based draw per restart:
        no pool to exhaust, no host work per step, nothing to redraw under a captured graph)."""
        from ..engine import EpisodeIO
        # what the cached start states / noise amplitudes were built from: a later change of stochastic_init or of the
        # environment's start pose (start_pos, start_rotation, start_joint_q, ...) must not be silently ignored by the fused
        # step while the torch reset() path honours it
        # (identity + in-place version counter of the tensors: no device work, no host sync -- this runs on every step)
        # (the key keeps a reference to each tensor: an id() alone could be recycled by a NEW tensor after the old one died)
        key = self._start_state_key()
        stale = getattr(self, "_pool", None) is not None and getattr(self, "_pool_key", None) != key
        if stale:
            self._check_pinned_pool(key)
        if getattr(self, "_pool", None) is None or stale:
            self._pool_key = key
            q0, qd0 = self._deterministic_start_state()
            if stale and self._pool[0].shape == q0.shape and self._pool[1].shape == qd0.shape:
                # in place: tensors handed out earlier (EpisodeIO objects of earlier steps) stay valid
                self._pool[0].copy_(q0)
                self._pool[1].copy_(qd0)
            else:
                self._pool = (q0, qd0)
            if getattr(self, "_reset_count", None) is None:
                self._reset_count = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            self._noise = None
            if bool(getattr(self, "stochastic_init", False)):
                n = self.reset_noise()
                if n is None:
                    raise NotImplementedError("%s: stochastic_init needs reset_noise() for the fused step" % type(self).__name__)
                nq, nqd, ang = n
                self._noise = (torch.as_tensor(nq, dtype=torch.float32).to(self.device).contiguous(),
                               torch.as_tensor(nqd, dtype=torch.float32).to(self.device).contiguous(), float(ang))
        if not self.progress_buf.is_contiguous():
            self.progress_buf = self.progress_buf.contiguous()
        nq, nqd, ang = self._noise if self._noise is not None else (None, None, 0.0)
        return EpisodeIO(self.progress_buf, self._pool[0], self._pool[1], self._reset_count, self.episode_length,
                         self.height_terminate, self.check_invalid, want_obs_before=not self.no_grad,
                         noise_q=nq, noise_qd=nqd, noise_angle=ang,
                         seed=self._philox_key())

    def _philox_key(self):
        """key of the in-kernel restart noise: the environment's seed mixed with the rank of the process, so that the shards of
        a multi-GPU job (same seed, local environment indices 0..n-1 on every rank) do not draw identical restart noise"""
        rank = int(os.environ.get("RANK", "0") or 0)
        return (int(self.seed) * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF

    def _step_fused(self, actions, spec):
        actions = actions.view((self.num_envs, self.num_actions))
        # (sanitize_grads: scrubbed inside the adjoint launch, see _spec; the unfused path below keeps the torch hooks)
        eng = self.model.engine()
        epi = self._episode_io()
        if self.no_grad:
            with torch.no_grad():
                q, qd, obs, rew, _, _, done = eng.env_forward(spec, self.state.joint_q.contiguous(),
                                                              self.state.joint_qd.contiguous(), actions.contiguous(),
                                                              float(self.sim_dt), self.sim_substeps,
                                                              self.MM_caching_frequency, False, epi)
            obs_before = None
        else:
            q, qd, obs, rew, obs_before, done = EnvStep.apply(eng, spec, epi, float(self.sim_dt), self.sim_substeps,
                                                              self.MM_caching_frequency, self.state.joint_q,
                                                              self.state.joint_qd, actions)
        st = df.State(act_like=self.model.joint_qd, model=self.model)
        st.joint_q, st.joint_qd = q, qd
        # (derived body transforms on request: of the returned joint_q -- a restarted environment has none in the checkpoint)
        self.state = st
        # self.actions (clipped / remapped, zero for restarted environments) is materialised on first use
        self._lazy_actions = (obs, spec.obs_actions, actions.detach(), done)
        self.obs_buf, self.rew_buf, self.reset_buf = obs, rew, done
        self.sim_time += self.sim_dt
        self.num_frames += 1
        if not self.no_grad:
            self.obs_buf_before_reset = obs_before
            self.extras = {"obs_before_reset": self.obs_buf_before_reset, "episode_end": self.termination_buf}
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    @property
    def actions(self):
        lazy = self.__dict__.get("_lazy_actions")
        if lazy is not None:
            obs, in_obs, raw, done = lazy
            if in_obs:
                self.__dict__["_actions"] = obs.detach()[:, self.num_observations - self.num_actions:]
            else:
                self.__dict__["_actions"] = torch.where(done.bool().unsqueeze(-1), torch.zeros_like(raw),
                                                        self.stored_actions(raw))
            self.__dict__["_lazy_actions"] = None
        return self.__dict__["_actions"]

    @actions.setter
    def actions(self, value):
        self.__dict__["_lazy_actions"] = None
        self.__dict__["_actions"] = value

    def _zeros_long(self):
        z = getattr(self, "_zl", None)
        if z is None:
            z = self._zl = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)
        return z

    def flag_resets(self):
        """sets reset_buf from obs_buf / progress_buf (the tail of the reference's calculateReward)"""
        self.reset_buf = torch.where(self.progress_buf > self.episode_length - 1, torch.ones_like(self.reset_buf),
                                     self.reset_buf)

    def _refresh_obs(self):
        """calculateObservations(); one kernel launch instead of ~25 torch ops when no gradient can flow through it"""
        spec = self._spec() if (self.fused and torch.device(self.device).type == "cuda") else None
        q, qd, a = self.state.joint_q, self.state.joint_qd, self.actions
        if spec is None or q.requires_grad or qd.requires_grad or a.requires_grad:
            self.calculateObservations()
        else:
            self.obs_buf, _ = self.model.engine().env_observe(spec, q.contiguous(), qd.contiguous(),
                                                              a.float().contiguous())
        return self.obs_buf

    # ---- hooks an environment implements ---------------------------------------------------------
    def apply_actions(self, actions):
        raise NotImplementedError

    def reset_state(self, env_ids):
        raise NotImplementedError

    def calculateObservations(self):
        raise NotImplementedError

    def calculateReward(self):
        raise NotImplementedError

    def render(self, mode="human"):
        pass

    # ---- the protocol ------------------------------------------------------------------------------
    def step(self, actions):
        spec = self._spec() if (self.fused and torch.device(self.device).type == "cuda") else None
        if spec is not None:
            return self._step_fused(actions, spec)
        actions = torch.clip(actions.view((self.num_envs, self.num_actions)), -1.0, 1.0)
        if self.sanitize_grads:
            def scrub(grad):
                return torch.nan_to_num(grad, 0.0, 0.0, 0.0)
            for t in (self.state.joint_q, self.state.joint_qd, actions):
                if t.requires_grad:
                    t.register_hook(scrub)
        self.apply_actions(actions)
        self.state = self.integrator.forward(self.model, self.state, self.sim_dt, self.sim_substeps,
                                             self.MM_caching_frequency)
        self.sim_time += self.sim_dt
        self.reset_buf = torch.zeros_like(self.reset_buf)
        self.progress_buf += 1
        self.num_frames += 1
        self.calculateObservations()
        self.calculateReward()
        env_ids = self.reset_buf.nonzero(as_tuple=False).squeeze(-1)
        if not self.no_grad:
            self.obs_buf_before_reset = self.obs_buf.clone()
            self.extras = {"obs_before_reset": self.obs_buf_before_reset, "episode_end": self.termination_buf}
        if len(env_ids) > 0:
            self.reset(env_ids)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset(self, env_ids=None, force_reset=True):
        if env_ids is None and force_reset:
            env_ids = torch.arange(self.num_envs, dtype=torch.long, device=self.device)
        if env_ids is not None:
            full = len(env_ids) == self.num_envs
            if full and self.fused and not getattr(self, "stochastic_init", False) and torch.device(self.device).type == "cuda":
                # every environment back to the (deterministic) start state: two copies from the start-state pool
                # instead of ~25 indexed writes
                epi = self._episode_io()
                pool_q, pool_qd = epi.reset_q, epi.reset_qd
                self.state.joint_q, self.state.joint_qd = pool_q[0].reshape(-1).clone(), pool_qd[0].reshape(-1).clone()
                self.actions = torch.zeros_like(self.actions)
            else:
                # fresh tensors: the old ones may be part of an autograd graph
                self.state.joint_q = self.state.joint_q.clone()
                self.state.joint_qd = self.state.joint_qd.clone()
                self.reset_state(env_ids)
            self.progress_buf[env_ids] = 0
            if len(env_ids) == self.num_envs:
                self._progress_hi = 0
            self._refresh_obs()
        return self.obs_buf

    def clear_grad(self, checkpoint=None):
        """Cuts the graph between the current state and everything before it."""
        with torch.no_grad():
            act = self.state.joint_act.clone() if self.keep_act_on_clear else None
            st = df.State(act_like=self.model.joint_qd, model=self.model)
            if checkpoint is None:
                # same result as restoring get_checkpoint(), without copying everything twice
                st.joint_q, st.joint_qd = self.state.joint_q.detach().clone(), self.state.joint_qd.detach().clone()
                self.actions = self.actions.detach().clone()
                self.progress_buf = self.progress_buf.clone()
            else:
                st.joint_q, st.joint_qd = checkpoint["joint_q"].clone(), checkpoint["joint_qd"].clone()
                self.actions = checkpoint["actions"].clone()
                if not self.fused and not torch.equal(self.progress_buf, checkpoint["progress_buf"]):
                    self._progress_hi = int(checkpoint["progress_buf"].max())
                self.progress_buf = checkpoint["progress_buf"].clone()
            self.state = st
            if act is not None:
                self.state.joint_act = act

    def detach_buffers(self):
        """drops every reference this object holds into an autograd graph (observation / reward buffers, extras, state):
        clear_grad() for the state plus the output buffers of the last step"""
        self.clear_grad()
        with torch.no_grad():
            self.obs_buf = self.obs_buf.detach()
            self.rew_buf = self.rew_buf.detach()
            if getattr(self, "obs_buf_before_reset", None) is not None:
                self.obs_buf_before_reset = self.obs_buf_before_reset.detach()
            self.extras = {}

    def initialize_trajectory(self):
        self.clear_grad()
        return self._refresh_obs()

    def _require_unit_quaternions(self, q):
        """The simulation path is defined on unit quaternions only (include/dsim.h, dsim_step_forward: 5e-3 error in qd at
        |q| = 1.001 against the reference's literal off-manifold formulas).  A state handed in from outside is checked here,
        where a host synchronisation is harmless; the step kernels check again asynchronously."""
        t = self.model.template()
        blocks = [int(cs) + (3 if int(ty) == df.JOINT_FREE else 0) for ty, cs in zip(t.joint_type, t.joint_q_start)
                  if int(ty) in (df.JOINT_FREE, df.JOINT_BALL)]
        if not blocks or q.numel() == 0:
            return
        with torch.no_grad():
            norms = torch.stack([q[:, b:b + 4].norm(dim=1) for b in blocks], dim=1)
            worst = float((norms - 1.0).abs().max())    # (nan compares false below and is reported too)
        if not worst <= 1e-4:
            raise ValueError("%s.reset_with_state: a quaternion block of joint_q is off the unit sphere by %.3g (> 1e-4); the "
                             "MI355X path is defined on unit quaternions only -- normalise the state first"
                             % (type(self).__name__, worst))

    def get_checkpoint(self):
        return {"joint_q": self.state.joint_q.clone(), "joint_qd": self.state.joint_qd.clone(),
                "actions": self.actions.clone(), "progress_buf": self.progress_buf.clone()}

    def get_state(self):
        return self.state.joint_q.clone(), self.state.joint_qd.clone()

    def reset_with_state(self, init_joint_q, init_joint_qd, env_ids=None, force_reset=True):
        if env_ids is None and force_reset:
            env_ids = torch.arange(self.num_envs, dtype=torch.long, device=self.device)
        if env_ids is not None:
            self._require_unit_quaternions(init_joint_q.view(-1, self.num_joint_q)[env_ids, :])
            self.state.joint_q = self.state.joint_q.clone()
            self.state.joint_qd = self.state.joint_qd.clone()
            self._q()[env_ids, :] = init_joint_q.view(-1, self.num_joint_q)[env_ids, :].clone()
            self._qd()[env_ids, :] = init_joint_qd.view(-1, self.num_joint_qd)[env_ids, :].clone()
            self.progress_buf[env_ids] = 0
            self._refresh_obs()
        return self.obs_buf
