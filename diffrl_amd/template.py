"""Single-articulation model template.

The reference replicates every model constant per environment (the asset is parsed N times,
envs/ant.py:102-113, and `ModelBuilder.finalize` uploads N copies, dflex/dflex/model.py:1646-1879).
Here ONE template describes the articulation; all N environments share it (it lives in LDS / the
scalar cache on the GPU) and only the state tensors are per-environment.

Field names follow the reference's `Model` attributes restricted to one articulation.
"""
from dataclasses import dataclass, field

import numpy as np

JOINT_PRISMATIC, JOINT_REVOLUTE, JOINT_BALL, JOINT_FIXED, JOINT_FREE = 0, 1, 2, 3, 4


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a.reshape(shape) if shape is not None else a


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


@dataclass
class ArticulationTemplate:
    joint_type: np.ndarray
    joint_parent: np.ndarray
    joint_q_start: np.ndarray
    joint_qd_start: np.ndarray
    joint_X_pj: np.ndarray
    joint_X_cm: np.ndarray
    joint_axis: np.ndarray
    body_I_m: np.ndarray
    joint_armature: np.ndarray
    joint_target: np.ndarray
    joint_target_ke: np.ndarray
    joint_target_kd: np.ndarray
    joint_limit_lower: np.ndarray
    joint_limit_upper: np.ndarray
    joint_limit_ke: np.ndarray
    joint_limit_kd: np.ndarray
    contact_body: np.ndarray
    contact_point: np.ndarray
    contact_dist: np.ndarray
    contact_material: np.ndarray  # [C,4] (ke, kd, kf, mu) gathered per contact
    muscle_start: np.ndarray
    muscle_links: np.ndarray
    muscle_points: np.ndarray
    gravity: np.ndarray
    joint_q0: np.ndarray = None   # rest pose written by the asset loader
    joint_qd0: np.ndarray = None
    extras: dict = field(default_factory=dict)

    def __post_init__(self):
        self.joint_type = _i32(self.joint_type)
        self.joint_parent = _i32(self.joint_parent)
        self.joint_q_start = _i32(self.joint_q_start)
        self.joint_qd_start = _i32(self.joint_qd_start)
        L = self.n_links
        self.joint_X_pj = _f32(self.joint_X_pj, (L, 7))
        self.joint_X_cm = _f32(self.joint_X_cm, (L, 7))
        self.joint_axis = _f32(self.joint_axis, (L, 3))
        self.body_I_m = _f32(self.body_I_m, (L, 6, 6))
        for k in ("joint_armature", "joint_target", "joint_target_ke", "joint_target_kd", "joint_limit_lower",
                  "joint_limit_upper", "joint_limit_ke", "joint_limit_kd", "contact_dist"):
            setattr(self, k, _f32(getattr(self, k)).reshape(-1))
        self.contact_body = _i32(self.contact_body).reshape(-1)
        C = self.contact_body.shape[0]
        self.contact_point = _f32(self.contact_point, (C, 3))
        self.contact_material = _f32(self.contact_material, (C, 4))
        self.muscle_start = _i32(self.muscle_start).reshape(-1)
        self.muscle_links = _i32(self.muscle_links).reshape(-1)
        self.muscle_points = _f32(self.muscle_points, (self.muscle_links.shape[0], 3))
        self.gravity = _f32(self.gravity, (3,))
        if self.joint_q0 is None:
            self.joint_q0 = np.zeros(self.n_q, np.float32)
        if self.joint_qd0 is None:
            self.joint_qd0 = np.zeros(self.n_qd, np.float32)
        self.joint_q0 = _f32(self.joint_q0).reshape(-1)
        self.joint_qd0 = _f32(self.joint_qd0).reshape(-1)
        self.validate()

    # sizes -------------------------------------------------------------------------------
    @property
    def n_links(self):
        return int(self.joint_type.shape[0])

    @property
    def n_q(self):
        return int(self.joint_q_start[-1])

    @property
    def n_qd(self):
        return int(self.joint_qd_start[-1])

    @property
    def n_contacts(self):
        return int(self.contact_body.shape[0])

    @property
    def n_muscles(self):
        return int(self.muscle_start.shape[0]) - 1

    @property
    def n_waypoints(self):
        return int(self.muscle_links.shape[0])

    def validate(self):
        L = self.n_links
        assert self.joint_q_start.shape[0] == L + 1 and self.joint_qd_start.shape[0] == L + 1
        assert np.all(self.joint_parent < np.arange(L)), "parents must precede children"
        assert self.joint_armature.shape[0] == self.n_qd
        assert self.joint_target.shape[0] == self.n_q
        assert self.muscle_start.shape[0] >= 1
        if self.n_contacts:
            assert self.contact_body.min() >= 0 and self.contact_body.max() < L
        if self.n_waypoints:
            assert self.muscle_links.min() >= 0 and self.muscle_links.max() < L

    # (de)serialisation -------------------------------------------------------------------
    _ARRAYS = ("joint_type joint_parent joint_q_start joint_qd_start joint_X_pj joint_X_cm joint_axis body_I_m "
               "joint_armature joint_target joint_target_ke joint_target_kd joint_limit_lower joint_limit_upper "
               "joint_limit_ke joint_limit_kd contact_body contact_point contact_dist contact_material "
               "muscle_start muscle_links muscle_points gravity joint_q0 joint_qd0").split()

    def to_dict(self):
        d = {k: getattr(self, k) for k in self._ARRAYS}
        for k, v in self.extras.items():
            d["extra_" + k] = np.asarray(v)
        return d

    def save(self, path):
        np.savez_compressed(path, **self.to_dict())

    @classmethod
    def from_dict(cls, d):
        kw = {k: d[k] for k in cls._ARRAYS}
        kw["extras"] = {k[6:]: d[k] for k in d.keys() if k.startswith("extra_")}
        return cls(**kw)

    @classmethod
    def load(cls, path):
        with np.load(path) as d:
            return cls.from_dict({k: d[k] for k in d.files})

    @classmethod
    def from_reference_dump(cls, d):
        """Builds a template from a `<env>_model.npz` golden (oracle/gen_golden.py:dump_model)."""
        cmat = np.asarray(d["shape_materials"], np.float32).reshape(-1, 4)
        cm = np.asarray(d["contact_material"], np.int64)
        contact_material = cmat[cm] if cm.size else np.zeros((0, 4), np.float32)
        extras = {}
        if "muscle_strengths" in d:
            extras["muscle_strengths"] = np.asarray(d["muscle_strengths"], np.float32)
        return cls(joint_type=d["joint_type"], joint_parent=d["joint_parent"], joint_q_start=d["joint_q_start"],
                   joint_qd_start=d["joint_qd_start"], joint_X_pj=d["joint_X_pj"], joint_X_cm=d["joint_X_cm"],
                   joint_axis=d["joint_axis"], body_I_m=d["body_I_m"], joint_armature=d["joint_armature"],
                   joint_target=d["joint_target"], joint_target_ke=d["joint_target_ke"],
                   joint_target_kd=d["joint_target_kd"], joint_limit_lower=d["joint_limit_lower"],
                   joint_limit_upper=d["joint_limit_upper"], joint_limit_ke=d["joint_limit_ke"],
                   joint_limit_kd=d["joint_limit_kd"], contact_body=d["contact_body"],
                   contact_point=d["contact_point"], contact_dist=d["contact_dist"],
                   contact_material=contact_material, muscle_start=d["muscle_start"],
                   muscle_links=d["muscle_links"], muscle_points=d["muscle_points"], gravity=d["gravity"],
                   joint_q0=d["joint_q0"], joint_qd0=d["joint_qd0"], extras=extras)
