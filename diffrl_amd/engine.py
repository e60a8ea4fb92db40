"""Torch-facing wrapper of the HIP library: device memory + stream plumbing and the autograd node.

`SimStep` replaces `dflex.sim.SimulateFunc` (dflex/dflex/sim.py:2086-2154): one autograd node per
env.step(); forward = one fused kernel launch over all substeps, backward = one fused adjoint launch.
"""
import ctypes as C

import torch

from . import capi
from .template import ArticulationTemplate


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    """Owns the device copy of one articulation template on one GPU."""

    def __init__(self, template: ArticulationTemplate, device, ckpt_mode=None, specialise=None):
        """ckpt_mode: "full" (default; $DIFFRL_AMD_CKPT overrides) -- the forward launch streams every intermediate the
        adjoint reads to HBM -- or "lean" -- (q, qd) per substep only, the adjoint recomputes (include/dsim.h).
        specialise: what a model that matches none of the compiled layout tables gets (it would run the generic kernels, about
        half the speed).  True / $DSIM_AUTO_SPECIALISE=1: a kernel set of its own, compiled with hipcc on first use and cached
        next to the library (diffrl_amd.specialise.ensure_library: about a minute, once per model and source version; the
        constructor waits for it).  "background" (the default when None and $DSIM_AUTO_SPECIALISE is unset or "background"): the
        cached set if there is one; otherwise this Engine runs the generic kernels while a detached child process compiles the set, and
        the next Engine of the model picks it up.  False / $DSIM_AUTO_SPECIALISE=0: nothing is compiled or swapped at run time.
        A failed build, a missing hipcc or a library that turns out not to hold the set: a warning, the generic kernels stay."""
        self.template = template
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise capi.DsimError("diffrl_amd runs on MI355X only (device=%s); there is no CPU path" % device)
        self._lib = capi.lib()
        if self.device.index is None:   # "cuda" means the current device; tensors report an explicit index
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._desc, self._keep = capi.make_desc(template)
        import os
        if specialise is None:
            v = os.environ.get("DSIM_AUTO_SPECIALISE", "background").lower()
            specialise = False if v in ("", "0", "off") else ("background" if v in ("background", "bg") else True)
            # (a developer override of the library -- A/B builds with a subset of the kernel sets -- is measured as it is: no
            # background swap behind its back)
            if specialise == "background" and os.environ.get("DSIM_LIB"):
                specialise = False
        h = self._create()
        if specialise and int(self._lib.dsim_model_variant(h)) == 0 and os.environ.get("DSIM_FORCE_GENERIC", "0") in ("", "0"):
            import warnings
            from . import specialise as sp
            if specialise == "background":
                path, _ = sp.ensure_library_background(template)   # None: compiling now (or no hipcc): generic kernels this time
            else:
                path = sp.ensure_library(template)   # None: hipcc missing or the build failed (warned about); the generic kernels stay
            # (the library already loaded reports variant 0 for this model: a path equal to it cannot help -- e.g. a product library
            # built with a DSIM_STATIC_VARIANTS subset, or a header regenerated without rebuilding)
            if path is not None and os.path.realpath(path) != os.path.realpath(getattr(self._lib, "_name", "") or ""):
                lib2 = capi.load(path)
                keep_lib, self._lib = self._lib, lib2
                try:
                    h2 = self._create()
                except capi.DsimError as ex:
                    h2, self._lib = None, keep_lib
                    warnings.warn("diffrl_amd: %s could not create this model (%s); it keeps the generic kernels" % (path, ex))
                if h2 is not None:
                    if int(lib2.dsim_model_variant(h2)) > 0:
                        keep_lib.dsim_model_destroy(h)
                        h = h2
                    else:   # keep the working generic handle, release the extra one
                        lib2.dsim_model_destroy(h2)
                        self._lib = keep_lib
                        warnings.warn("diffrl_amd: %s was built for this model but does not match it; it keeps the generic kernels"
                                      % path)
            elif path is not None:
                warnings.warn("diffrl_amd: the loaded library %s lists a kernel set for this model but was built without it; "
                              "it keeps the generic kernels" % path)
        self._h = h
        self.ckpt_mode = (ckpt_mode or os.environ.get("DIFFRL_AMD_CKPT", "full")).lower()
        if self.ckpt_mode not in ("full", "lean"):
            raise capi.DsimError("ckpt_mode must be 'full' or 'lean'")
        with torch.cuda.device(self.device):   # (re-evaluates the occupancy of the helper-wave kernels on the model's device)
            self._ck(self._lib.dsim_model_set_ckpt_mode(h, capi.CKPT_LEAN if self.ckpt_mode == "lean" else capi.CKPT_FULL))
        assert int(self._lib.dsim_model_device(h)) == self.device.index
        self.variant = int(self._lib.dsim_model_variant(h))  # 0 = generic kernels, > 0 = specialised for this model
        self.n_q, self.n_qd, self.n_muscles = template.n_q, template.n_qd, template.n_muscles
        # joint_q ENTERING the last substep of the most recent forward() with gradients on ([n_envs, n_q], a copy: n_q floats per
        # environment -- never a reference to the checkpoint itself, which is hundreds of MB for the humanoid and must die with
        # its autograd node): what State.body_X_sc derives the reference's lagging transforms from (dflex/sim.py)
        self.last_q_in = None

    def _ck(self, rc):
        capi.check(rc, self._lib)

    def _create(self):
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            self._ck(self._lib.dsim_model_create(C.byref(self._desc), C.byref(h)))
        return h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.dsim_model_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def status(self):
        """Raises DsimError if a forward launch since the last report was handed a non-unit quaternion (include/dsim.h:
        the path is defined on unit quaternions only).  Host-side read of two mapped words; synchronise first for a
        definitive answer about launches still in flight."""
        self._ck(self._lib.dsim_model_status(self._h, None))

    def body_transforms(self, q):
        """(X_sc, X_sm), each [n_envs * n_links, 7]: link frames and centre-of-mass frames in the world for the joint
        coordinates q -- the reference's State.body_X_sc / body_X_sm (dflex/dflex/model.py:338-392)."""
        q = q.detach().contiguous()
        self._check(q, self.n_q, "joint_q")
        n = q.numel() // self.n_q
        L = self.template.n_links
        xsc = torch.empty((n * L, 7), dtype=torch.float32, device=self.device)
        xsm = torch.empty((n * L, 7), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._ck(self._lib.dsim_body_transforms(self._h, n, _ptr(q), _ptr(xsc), _ptr(xsm), st))
        return xsc, xsm

    def last_substep_q(self, ckpt, substeps):
        """joint_q ENTERING the last substep of the step that wrote `ckpt` (the head of that substep's checkpoint row): what the
        reference's eval_rigid_fk saw when it filled the returned State's body_X_sc (sim.py:2316-2601)."""
        row = int(self._lib.dsim_ckpt_floats_mm(self._h, 2, 1 << 30)) - int(self._lib.dsim_ckpt_floats_mm(self._h, 1, 1 << 30))
        return ckpt[:, (substeps - 1) * row:(substeps - 1) * row + self.n_q].contiguous()

    def _alloc_ckpt(self, n, substeps, mm_freq):
        """[n][dsim_ckpt_floats_mm]: per substep the saved forward block (starts with q, qd), then the H^-1 per group"""
        words = int(self._lib.dsim_ckpt_floats_mm(self._h, substeps, mm_freq))
        return torch.empty((n, words), dtype=torch.float32, device=self.device)

    def _check(self, t, cols, name):
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            raise capi.DsimError("%s must be a contiguous float32 tensor on %s" % (name, self.device))
        if t.numel() % max(cols, 1) != 0:
            raise capi.DsimError("%s has %d elements, not a multiple of %d" % (name, t.numel(), cols))

    def forward(self, q, qd, act, mact, dt, substeps, mm_freq, need_ckpt, keep_q_in=False):
        """keep_q_in: also copy joint_q entering the last substep out of the checkpoint (-> self.last_q_in; one small copy kernel,
        asked for by the operator boundary only: SimStep / SemiImplicitIntegrator.forward)"""
        self._check(q, self.n_q, "joint_q")
        self._check(qd, self.n_qd, "joint_qd")
        self._check(act, self.n_qd, "joint_act")
        n = q.numel() // self.n_q
        if qd.numel() != n * self.n_qd or act.numel() != n * self.n_qd:
            raise capi.DsimError("state tensors disagree on the number of environments")
        if self.n_muscles:
            self._check(mact, self.n_muscles, "muscle_activation")
            if mact.numel() != n * self.n_muscles:
                raise capi.DsimError("muscle_activation has the wrong size")
        q_out = torch.empty_like(q)
        qd_out = torch.empty_like(qd)
        ckpt = None
        if need_ckpt:
            ckpt = self._alloc_ckpt(n, substeps, mm_freq)
        self.last_q_in = None
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._ck(self._lib.dsim_step_forward(self._h, n, _ptr(q), _ptr(qd), _ptr(act),
                                                   _ptr(mact) if self.n_muscles else None, C.c_float(dt), substeps,
                                                   mm_freq, _ptr(q_out), _ptr(qd_out), _ptr(ckpt), st))
        if ckpt is not None and keep_q_in:
            self.last_q_in = self.last_substep_q(ckpt, substeps)
        return q_out, qd_out, ckpt

    def _check_ckpt(self, ckpt, substeps, mm_freq):
        """a checkpoint must be consumed with the geometry (substeps, mass-matrix frequency, checkpoint mode) it was written
        with: the row stride differs otherwise and the adjoint launch would read other rows' words"""
        words = int(self._lib.dsim_ckpt_floats_mm(self._h, substeps, mm_freq))
        if (ckpt.dim() != 2 or ckpt.shape[1] != words or ckpt.device != self.device or ckpt.dtype != torch.float32
                or not ckpt.is_contiguous()):   # (a strided view, e.g. ckpt[::2], has the right shape and the wrong row stride)
            raise capi.DsimError("checkpoint of shape %s does not match this model / step geometry (%d floats per environment "
                                 "in '%s' mode on %s, contiguous rows)" % (tuple(ckpt.shape), words, self.ckpt_mode, self.device))

    def backward(self, ckpt, act, mact, dt, substeps, mm_freq, gq_out, gqd_out, literal=False):
        """literal: dsim_step_backward_literal -- the quaternion blocks of the returned gq carry the component along the quaternion
        that the reference's literal adjoint has (one more small launch; default: the wrench form, no such component)"""
        self._check_ckpt(ckpt, substeps, mm_freq)
        n = ckpt.shape[0]
        gq_out = gq_out.contiguous()
        gqd_out = gqd_out.contiguous()
        gq = torch.empty(n * self.n_q, dtype=torch.float32, device=self.device)
        gqd = torch.empty(n * self.n_qd, dtype=torch.float32, device=self.device)
        gact = torch.empty(n * self.n_qd, dtype=torch.float32, device=self.device)
        gm = torch.empty(n * self.n_muscles, dtype=torch.float32, device=self.device) if self.n_muscles else None
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            if literal:
                scratch = torch.empty((n, int(self._lib.dsim_literal_scratch_floats(self._h))), dtype=torch.float32, device=self.device)
                self._ck(self._lib.dsim_step_backward_literal(self._h, n, _ptr(ckpt), _ptr(act),
                                                                _ptr(mact) if self.n_muscles else None, C.c_float(dt), substeps,
                                                                mm_freq, _ptr(gq_out), _ptr(gqd_out), _ptr(gq), _ptr(gqd),
                                                                _ptr(gact), _ptr(gm), _ptr(scratch), st))
            else:
                self._ck(self._lib.dsim_step_backward(self._h, n, _ptr(ckpt), _ptr(act),
                                                        _ptr(mact) if self.n_muscles else None, C.c_float(dt), substeps,
                                                        mm_freq, _ptr(gq_out), _ptr(gqd_out), _ptr(gq), _ptr(gqd),
                                                        _ptr(gact), _ptr(gm), st))
        return gq, gqd, gact, gm


    # ---- fused environment surface ------------------------------------------------------------------
    def env_forward(self, spec, q, qd, actions, dt, substeps, mm_freq, need_ckpt, episode=None):
        """-> (q_out, qd_out, obs, rew, ckpt[, obs_before_reset, done] when `episode` (an EpisodeIO) is given)"""
        self._check(q, self.n_q, "joint_q")
        self._check(qd, self.n_qd, "joint_qd")
        self._check(actions, spec.n_act, "actions")
        n = q.numel() // self.n_q
        if qd.numel() != n * self.n_qd or actions.numel() != n * spec.n_act:
            raise capi.DsimError("state / action tensors disagree on the number of environments")
        q_out, qd_out = torch.empty_like(q), torch.empty_like(qd)
        obs = torch.empty((n, spec.n_obs), dtype=torch.float32, device=self.device)
        rew = torch.empty(n, dtype=torch.float32, device=self.device)
        ckpt = self._alloc_ckpt(n, substeps, mm_freq) if need_ckpt else None
        ep, extra = None, ()
        if episode is not None:
            ep, extra = episode.bind(self, n, spec.n_obs)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._ck(self._lib.dsim_env_step_forward(self._h, C.byref(spec), n, _ptr(q), _ptr(qd), _ptr(actions),
                                                       C.c_float(dt), substeps, mm_freq, _ptr(q_out), _ptr(qd_out),
                                                       _ptr(obs), _ptr(rew), _ptr(ckpt),
                                                       C.byref(ep) if ep is not None else None, st))
        return (q_out, qd_out, obs, rew, ckpt) + extra

    def env_backward(self, spec, ckpt, actions, dt, substeps, mm_freq, gq_out, gqd_out, gobs, grew, gobs_before=None):
        """any cotangent may be None (= zeros: no fill kernels are launched for unused outputs)"""
        self._check_ckpt(ckpt, substeps, mm_freq)
        n = ckpt.shape[0]
        gq = torch.empty(n * self.n_q, dtype=torch.float32, device=self.device)
        gqd = torch.empty(n * self.n_qd, dtype=torch.float32, device=self.device)
        ga = torch.empty((n, spec.n_act), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._ck(self._lib.dsim_env_step_backward(self._h, C.byref(spec), n, _ptr(ckpt), _ptr(actions),
                                                        C.c_float(dt), substeps, mm_freq, _ptr(gq_out), _ptr(gqd_out),
                                                        _ptr(gobs), _ptr(grew), _ptr(gobs_before), _ptr(gq), _ptr(gqd),
                                                        _ptr(ga), st))
        return gq, gqd, ga

    def env_observe(self, spec, q, qd, stored_actions):
        n = q.numel() // self.n_q
        obs = torch.empty((n, spec.n_obs), dtype=torch.float32, device=self.device)
        rew = torch.empty(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._ck(self._lib.dsim_env_observe(self._h, C.byref(spec), n, _ptr(q), _ptr(qd), _ptr(stored_actions),
                                                  _ptr(obs), _ptr(rew), st))
        return obs, rew


class EpisodeIO:
    """Device buffers of the in-kernel episode bookkeeping (`dsim_episode`, include/dsim.h): progress_buf, the pool of
    start states finished environments restart from, and the termination rules."""

    def __init__(self, progress, reset_q, reset_qd, reset_count, episode_length, height_terminate, check_invalid,
                 want_obs_before, noise_q=None, noise_qd=None, noise_angle=0.0, seed=0):
        self.progress, self.reset_q, self.reset_qd, self.reset_count = progress, reset_q, reset_qd, reset_count
        self.episode_length, self.height_terminate, self.check_invalid = episode_length, height_terminate, check_invalid
        self.want_obs_before = want_obs_before
        # in-kernel stochastic restart: per-coordinate noise amplitudes [n_q] / [n_qd] (device, float32) or None
        self.noise_q, self.noise_qd, self.noise_angle, self.seed = noise_q, noise_qd, float(noise_angle), int(seed)

    def bind(self, engine, n, n_obs):
        dev = engine.device
        for t, dt_, name in ((self.progress, torch.int64, "progress"), (self.reset_count, torch.int32, "reset_count")):
            if t.device != dev or t.dtype != dt_ or t.numel() != n or not t.is_contiguous():
                raise capi.DsimError("episode.%s must be a contiguous %s tensor of %d elements on %s" % (name, dt_, n, dev))
        k = self.reset_q.shape[0]
        for t, cols, name in ((self.reset_q, engine.n_q, "reset_q"), (self.reset_qd, engine.n_qd, "reset_qd")):
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != k * n * cols:
                raise capi.DsimError("episode.%s must be a contiguous float32 [pool][n_envs][%d] tensor on %s" % (name, cols, dev))
        done = torch.empty(n, dtype=torch.int64, device=dev)
        obs_before = torch.empty((n, n_obs), dtype=torch.float32, device=dev) if self.want_obs_before else None
        ep = capi.Episode()
        ep.progress, ep.done = self.progress.data_ptr(), done.data_ptr()
        ep.obs_before_reset = obs_before.data_ptr() if obs_before is not None else None
        ep.reset_q, ep.reset_qd, ep.reset_count = self.reset_q.data_ptr(), self.reset_qd.data_ptr(), self.reset_count.data_ptr()
        ep.reset_pool, ep.episode_length = int(k), int(self.episode_length)
        ep.height_terminate, ep.check_invalid = int(bool(self.height_terminate)), int(bool(self.check_invalid))
        for t, cols, name in ((self.noise_q, engine.n_q, "noise_q"), (self.noise_qd, engine.n_qd, "noise_qd")):
            if t is not None and (t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != cols):
                raise capi.DsimError("episode.%s must be a contiguous float32 [%d] tensor on %s" % (name, cols, dev))
        ep.noise_q = self.noise_q.data_ptr() if self.noise_q is not None else None
        ep.noise_qd = self.noise_qd.data_ptr() if self.noise_qd is not None else None
        ep.noise_angle, ep.seed = self.noise_angle, self.seed & 0xFFFFFFFFFFFFFFFF
        return ep, (obs_before, done)


class EnvStep(torch.autograd.Function):
    """Fused env.step(): (joint_q, joint_qd, actions) -> (joint_q', joint_qd', obs, rew); ONE launch each way.
    With an EpisodeIO the launch also does the episode bookkeeping and the outputs gain (obs_before_reset, done)."""

    @staticmethod
    def forward(ctx, engine, spec, episode, dt, substeps, mm_freq, q, qd, actions):
        q, qd, actions = q.contiguous(), qd.contiguous(), actions.contiguous()
        need = q.requires_grad or qd.requires_grad or actions.requires_grad
        out = engine.env_forward(spec, q.detach(), qd.detach(), actions.detach(), dt, substeps, mm_freq, need, episode)
        q_out, qd_out, obs, rew, ckpt = out[:5]
        ctx.engine, ctx.spec, ctx.dt, ctx.substeps, ctx.mm_freq = engine, spec, dt, substeps, mm_freq
        ctx.shapes = (q.shape, qd.shape, actions.shape)
        ctx.set_materialize_grads(False)
        if need:
            ctx.save_for_backward(ckpt, actions.detach())
        res = (q_out.view(q.shape), qd_out.view(qd.shape), obs, rew)
        if episode is not None:
            obs_before, done = out[5], out[6]
            ctx.mark_non_differentiable(done)
            res = res + ((obs_before if obs_before is not None else obs.new_empty(0)), done)
        return res

    @staticmethod
    def backward(ctx, gq_out, gqd_out, gobs, grew, gobs_before=None, gdone=None):
        ckpt, actions = ctx.saved_tensors
        c = lambda g: g.contiguous() if g is not None else None  # noqa: E731
        if gobs_before is not None and gobs_before.numel() == 0:
            gobs_before = None
        gq, gqd, ga = ctx.engine.env_backward(ctx.spec, ckpt, actions, ctx.dt, ctx.substeps, ctx.mm_freq, c(gq_out),
                                              c(gqd_out), c(gobs), c(grew), c(gobs_before))
        return None, None, None, None, None, None, gq.view(ctx.shapes[0]), gqd.view(ctx.shapes[1]), ga.view(ctx.shapes[2])


class SimStep(torch.autograd.Function):
    """(joint_q, joint_qd, joint_act, muscle_activation) -> (joint_q', joint_qd') for one env.step()."""

    @staticmethod
    def forward(ctx, engine, dt, substeps, mm_freq, q, qd, act, mact):
        q, qd, act = q.contiguous(), qd.contiguous(), act.contiguous()
        mact = mact.contiguous() if mact is not None else None
        need = any(t is not None and t.requires_grad for t in (q, qd, act, mact))
        q_out, qd_out, ckpt = engine.forward(q.detach(), qd.detach(), act.detach(),
                                             mact.detach() if mact is not None else None, dt, substeps, mm_freq, need,
                                             keep_q_in=True)
        ctx.engine, ctx.dt, ctx.substeps, ctx.mm_freq = engine, dt, substeps, mm_freq
        ctx.has_mact = mact is not None
        ctx.shapes = (q.shape, qd.shape, act.shape, mact.shape if mact is not None else None)
        if need:
            ctx.save_for_backward(ckpt, act.detach(), mact.detach() if mact is not None else act.new_empty(0))
        return q_out.view(q.shape), qd_out.view(qd.shape)

    @staticmethod
    def backward(ctx, gq_out, gqd_out):
        ckpt, act, mact = ctx.saved_tensors
        e = ctx.engine
        if gq_out is None:
            gq_out = torch.zeros(ctx.shapes[0], dtype=torch.float32, device=ckpt.device)
        if gqd_out is None:
            gqd_out = torch.zeros(ctx.shapes[1], dtype=torch.float32, device=ckpt.device)
        from .dflex import config
        gq, gqd, gact, gm = e.backward(ckpt, act, mact if ctx.has_mact else None, ctx.dt, ctx.substeps, ctx.mm_freq,
                                       gq_out, gqd_out, literal=config.literal_quat_grad)
        return (None, None, None, None, gq.view(ctx.shapes[0]), gqd.view(ctx.shapes[1]), gact.view(ctx.shapes[2]),
                gm.view(ctx.shapes[3]) if ctx.has_mact else None)
