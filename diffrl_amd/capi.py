"""ctypes binding of include/dsim.h (the C-ABI shared library `csrc/libdsim_hip.so`).

This is the ONLY compute path of the package: there is no CPU / eager fallback.  If the HIP
library is missing or fails to load, `lib()` raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSIM_LIB") or os.path.join(_HERE, "csrc", "libdsim_hip.so")  # DSIM_LIB: developer override (A/B builds)

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)


class ModelDesc(C.Structure):
    """Mirror of `dsim_model_desc` (include/dsim.h)."""
    _fields_ = [
        ("n_links", C.c_int32), ("n_q", C.c_int32), ("n_qd", C.c_int32), ("n_contacts", C.c_int32),
        ("n_muscles", C.c_int32), ("n_waypoints", C.c_int32),
        ("joint_type", _i32p), ("joint_parent", _i32p), ("joint_q_start", _i32p), ("joint_qd_start", _i32p),
        ("joint_X_pj", _f32p), ("joint_X_cm", _f32p), ("joint_axis", _f32p), ("body_I_m", _f32p),
        ("joint_armature", _f32p), ("joint_target", _f32p), ("joint_target_ke", _f32p), ("joint_target_kd", _f32p),
        ("joint_limit_lower", _f32p), ("joint_limit_upper", _f32p), ("joint_limit_ke", _f32p),
        ("joint_limit_kd", _f32p),
        ("contact_body", _i32p), ("contact_point", _f32p), ("contact_dist", _f32p), ("contact_material", _f32p),
        ("muscle_start", _i32p), ("muscle_links", _i32p), ("muscle_points", _f32p),
        ("gravity", C.c_float * 3),
    ]


class EnvSpec(C.Structure):
    """Mirror of `dsim_env_spec` (include/dsim.h)."""
    _fields_ = [
        ("kind", C.c_int32), ("rew_kind", C.c_int32), ("n_act", C.c_int32), ("n_obs", C.c_int32),
        ("act_offset", C.c_int32), ("act_muscle", C.c_int32), ("obs_actions", C.c_int32), ("sanitize_grads", C.c_int32),
        ("inv_start_rot", C.c_float * 4), ("target_x", C.c_float), ("target_z", C.c_float),
        ("termination_height", C.c_float), ("termination_tolerance", C.c_float), ("height_rew_scale", C.c_float),
        ("action_penalty", C.c_float), ("joint_vel_obs_scaling", C.c_float), ("cartpole_penalties", C.c_float * 4),
        ("act_scale", C.c_void_p),
    ]


class Episode(C.Structure):
    """Mirror of `dsim_episode` (include/dsim.h): episode bookkeeping fused into the step kernel."""
    _fields_ = [
        ("progress", C.c_void_p), ("done", C.c_void_p), ("obs_before_reset", C.c_void_p), ("reset_q", C.c_void_p),
        ("reset_qd", C.c_void_p), ("reset_count", C.c_void_p), ("reset_pool", C.c_int32),
        ("episode_length", C.c_int32), ("height_terminate", C.c_int32), ("check_invalid", C.c_int32),
        ("noise_q", C.c_void_p), ("noise_qd", C.c_void_p), ("noise_angle", C.c_float), ("seed", C.c_uint64),
    ]


ENV_LOCOMOTION, ENV_CARTPOLE, ENV_PLANAR = 1, 2, 3
CKPT_FULL, CKPT_LEAN = 0, 1
OK, ERR_INVALID, ERR_HIP, ERR_LIMIT = 0, -1, -2, -3   # include/dsim.h: DSIM_OK, DSIM_ERR_*
REW_ANT, REW_HUMANOID, REW_SNU, REW_CARTPOLE, REW_HOPPER, REW_CHEETAH = 0, 1, 2, 3, 4, 5


def make_env_spec(kind, rew_kind, n_act, n_obs, act_scale_ptr, act_offset=0, act_muscle=False, obs_actions=False,
                  inv_start_rot=(0.0, 0.0, 0.0, 1.0), target_xz=(0.0, 0.0), termination_height=0.0,
                  termination_tolerance=0.0, height_rew_scale=0.0, action_penalty=0.0, joint_vel_obs_scaling=0.1,
                  cartpole_penalties=(0.0, 0.0, 0.0, 0.0)):
    e = EnvSpec()
    e.kind, e.rew_kind, e.n_act, e.n_obs = kind, rew_kind, n_act, n_obs
    e.act_offset, e.act_muscle, e.obs_actions = act_offset, int(bool(act_muscle)), int(bool(obs_actions))
    for k in range(4):
        e.inv_start_rot[k] = float(inv_start_rot[k])
        e.cartpole_penalties[k] = float(cartpole_penalties[k])
    e.target_x, e.target_z = float(target_xz[0]), float(target_xz[1])
    e.termination_height, e.termination_tolerance = float(termination_height), float(termination_tolerance)
    e.height_rew_scale, e.action_penalty = float(height_rew_scale), float(action_penalty)
    e.joint_vel_obs_scaling = float(joint_vel_obs_scaling)
    e.act_scale = act_scale_ptr
    return e


def make_desc(t):
    """ArticulationTemplate -> (ModelDesc, keepalive list).  Arrays are borrowed: keep `t` alive."""
    d = ModelDesc()
    d.n_links, d.n_q, d.n_qd = t.n_links, t.n_q, t.n_qd
    d.n_contacts, d.n_muscles, d.n_waypoints = t.n_contacts, t.n_muscles, t.n_waypoints
    keep = []

    def fp(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        keep.append(a)
        return a.ctypes.data_as(_f32p)

    def ip(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return a.ctypes.data_as(_i32p)

    for name in ("joint_type", "joint_parent", "joint_q_start", "joint_qd_start", "contact_body", "muscle_start",
                 "muscle_links"):
        setattr(d, name, ip(getattr(t, name)))
    for name in ("joint_X_pj", "joint_X_cm", "joint_axis", "body_I_m", "joint_armature", "joint_target",
                 "joint_target_ke", "joint_target_kd", "joint_limit_lower", "joint_limit_upper", "joint_limit_ke",
                 "joint_limit_kd", "contact_point", "contact_dist", "contact_material", "muscle_points"):
        setattr(d, name, fp(getattr(t, name)))
    for k in range(3):
        d.gravity[k] = float(t.gravity[k])
    return d, keep


class DsimError(RuntimeError):
    pass


_libs = {}
EXPECTED_ABI = 107   # dsim_version() of the library this binding was written against (argument lists of include/dsim.h)


def lib():
    """Loads libdsim_hip.so; raises if it is absent (no fallback by design)."""
    return load(LIB_PATH)


def load(path):
    """Binding of the library at `path` (libdsim_hip.so, or a library of the same sources built for a user model by
    diffrl_amd.specialise); one handle per path."""
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise DsimError("HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU fallback)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.dsim_last_error.restype = C.c_char_p
    L.dsim_version.restype = C.c_int
    if int(L.dsim_version()) != EXPECTED_ABI:
        raise DsimError("%s reports ABI version %d, this binding expects %d: stale build or a DSIM_LIB override built "
                        "against other argument lists; rebuild (DSIM_FORCE_REBUILD=1 python __graft_entry__.py)"
                        % (path, int(L.dsim_version()), EXPECTED_ABI))
    L.dsim_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(vp)]
    L.dsim_model_destroy.argtypes = [vp]
    L.dsim_model_variant.argtypes = [vp]
    L.dsim_model_variant.restype = C.c_int
    L.dsim_model_device.argtypes = [vp]
    L.dsim_model_device.restype = C.c_int
    L.dsim_ckpt_floats.argtypes = [vp, C.c_int]
    L.dsim_ckpt_floats.restype = C.c_int64
    L.dsim_ckpt_floats_mm.argtypes = [vp, C.c_int, C.c_int]
    L.dsim_ckpt_floats_mm.restype = C.c_int64
    L.dsim_model_set_ckpt_mode.argtypes = [vp, C.c_int]
    L.dsim_model_set_ckpt_mode.restype = C.c_int
    L.dsim_step_forward.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp]
    L.dsim_step_backward.argtypes = [vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp,
                                     vp]
    L.dsim_step_backward_literal.argtypes = [vp, C.c_int, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.dsim_step_backward_literal.restype = C.c_int
    L.dsim_literal_scratch_floats.argtypes = [vp]
    L.dsim_literal_scratch_floats.restype = C.c_int64
    ep = C.POINTER(EnvSpec)
    L.dsim_env_step_forward.argtypes = [vp, ep, C.c_int, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                        C.POINTER(Episode), vp]
    L.dsim_env_step_backward.argtypes = [vp, ep, C.c_int, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                         vp, vp, vp, vp]
    L.dsim_env_observe.argtypes = [vp, ep, C.c_int, vp, vp, vp, vp, vp, vp]
    L.dsim_model_status.argtypes = [vp, C.POINTER(C.c_int)]
    L.dsim_body_transforms.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    for fn in (L.dsim_model_create, L.dsim_model_destroy, L.dsim_step_forward, L.dsim_step_backward,
               L.dsim_env_step_forward, L.dsim_env_step_backward, L.dsim_env_observe, L.dsim_model_status,
               L.dsim_body_transforms):
        fn.restype = C.c_int
    _libs[path] = L
    return L


def check(rc, L=None):
    """L: the library the call went to (default: the product library)"""
    if rc != 0:
        raise DsimError("dsim error %d: %s" % (rc, (L or lib()).dsim_last_error().decode()))


EXPORTS = ("dsim_last_error", "dsim_version", "dsim_model_create", "dsim_model_destroy", "dsim_model_variant", "dsim_model_device",
           "dsim_ckpt_floats", "dsim_ckpt_floats_mm", "dsim_model_set_ckpt_mode",
           "dsim_step_forward", "dsim_step_backward", "dsim_step_backward_literal", "dsim_literal_scratch_floats", "dsim_env_step_forward", "dsim_env_step_backward",
           "dsim_env_observe", "dsim_model_status", "dsim_body_transforms")
