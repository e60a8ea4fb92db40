"""ModelBuilder / Model / State: the articulation subset of dflex/dflex/model.py, MI355X-first.

The reference builds N copies of every constant (one per environment).  Here a builder describes
articulations as usual (same `add_link` / `add_shape_*` / `add_muscle` calls), but `finalize` keeps a
single `ArticulationTemplate` plus a replica count: all environments share the template on the GPU.
Builders that were filled with N identical articulations (as the reference's env constructors do) are
accepted and collapsed; heterogeneous scenes are rejected (out of scope).
"""
import math

import numpy as np
import torch

from ..template import (JOINT_BALL, JOINT_FIXED, JOINT_FREE, JOINT_PRISMATIC, JOINT_REVOLUTE,
                        ArticulationTemplate)
from . import util as U

GEO_SPHERE, GEO_BOX, GEO_CAPSULE, GEO_MESH, GEO_SDF, GEO_PLANE, GEO_NONE = 0, 1, 2, 3, 4, 5, 6

_COORDS = {JOINT_PRISMATIC: 1, JOINT_REVOLUTE: 1, JOINT_BALL: 4, JOINT_FIXED: 0, JOINT_FREE: 7}
_DOFS = {JOINT_PRISMATIC: 1, JOINT_REVOLUTE: 1, JOINT_BALL: 3, JOINT_FIXED: 0, JOINT_FREE: 6}


def _solid_mass_props(kind, scale, density):
    """(mass, 3x3 inertia about the shape's own frame); closed forms as dflex/dflex/model.py:1530-1627."""
    if density == 0:
        return 0.0, np.zeros((3, 3))
    if kind == GEO_SPHERE:
        r = scale[0]
        m = density * (4.0 / 3.0 * math.pi * r * r * r)
        i = 2.0 / 5.0 * m * r * r
        return m, np.diag([i, i, i])
    if kind == GEO_BOX:
        w, h, d = scale[0] * 2.0, scale[1] * 2.0, scale[2] * 2.0
        m = density * (w * h * d)
        return m, np.diag([1.0 / 12.0 * m * (h * h + d * d), 1.0 / 12.0 * m * (w * w + d * d),
                           1.0 / 12.0 * m * (w * w + h * h)])
    if kind == GEO_CAPSULE:
        r, l = scale[0], scale[1] * 2.0
        ms = density * (4.0 / 3.0) * math.pi * r * r * r
        mc = density * math.pi * r * r * l
        ia = mc * (0.25 * r * r + (1.0 / 12.0) * l * l) + ms * (0.4 * r * r + 0.375 * r * l + 0.25 * l * l)
        ib = (mc * 0.5 + ms * 0.4) * r * r
        return ms + mc, np.diag([ib, ia, ia])
    raise NotImplementedError("shape type %d (meshes/SDFs are outside the articulated ground-contact path)" % kind)


class ModelBuilder:
    """Collects links, shapes and muscles of one or more (identical) articulations."""

    def __init__(self):
        self.articulation_start = []
        self.joint_type, self.joint_parent, self.joint_axis, self.joint_X_pj = [], [], [], []
        self.joint_q_start, self.joint_qd_start = [], []
        self.joint_q, self.joint_qd, self.joint_target = [], [], []
        self.joint_armature, self.joint_limit_lower, self.joint_limit_upper = [], [], []
        self.joint_target_ke, self.joint_target_kd, self.joint_limit_ke, self.joint_limit_kd = [], [], [], []
        self.body_mass, self.body_inertia, self.body_com = [], [], []
        self.shape_body, self.shape_transform, self.shape_geo_type, self.shape_geo_scale = [], [], [], []
        self.shape_materials = []
        self.muscle_start, self.muscle_params, self.muscle_activation = [], [], []
        self.muscle_links, self.muscle_points = [], []

    # ---- articulation description (same call signatures as the reference's builder) ------------
    def add_articulation(self):
        self.articulation_start.append(len(self.joint_type))
        return len(self.articulation_start) - 1

    def add_link(self, parent, X_pj, axis, type, armature=0.01, stiffness=0.0, damping=0.0, limit_lower=-1.e+3,
                 limit_upper=1.e+3, limit_ke=100.0, limit_kd=10.0, com=np.zeros(3), I_m=np.zeros((3, 3)), m=0.0):
        nq, nd = _COORDS[type], _DOFS[type]
        self.joint_type.append(type)
        self.joint_parent.append(parent)
        self.joint_axis.append(np.array(axis, dtype=np.float64))
        self.joint_X_pj.append((np.array(X_pj[0], dtype=np.float64), np.array(X_pj[1], dtype=np.float64)))
        self.joint_target_ke.append(stiffness)
        self.joint_target_kd.append(damping)
        self.joint_limit_ke.append(limit_ke)
        self.joint_limit_kd.append(limit_kd)
        self.joint_q_start.append(len(self.joint_q))
        self.joint_qd_start.append(len(self.joint_qd))
        if type in (JOINT_PRISMATIC, JOINT_REVOLUTE):
            self.joint_q += [0.0]
            self.joint_limit_lower += [limit_lower]
            self.joint_limit_upper += [limit_upper]
            self.joint_armature += [armature]
        elif type == JOINT_BALL:
            self.joint_q += [0.0, 0.0, 0.0, 1.0]
            self.joint_limit_lower += [limit_lower] * 3 + [0.0]
            self.joint_limit_upper += [limit_upper] * 3 + [0.0]
            self.joint_armature += [armature] * 3
        elif type == JOINT_FREE:
            self.joint_q += [0.0] * 6 + [1.0]
            self.joint_limit_lower += [0.0] * 7
            self.joint_limit_upper += [0.0] * 7
            self.joint_armature += [0.0] * 6  # the free joint carries no armature
        self.joint_qd += [0.0] * nd
        self.joint_target += [0.0] * nq
        self.body_inertia.append(np.zeros((3, 3)))
        self.body_mass.append(0.0)
        self.body_com.append(np.zeros(3))
        return len(self.joint_type) - 1

    def add_muscle(self, links, positions, f0, lm, lt, lmax, pen):
        self.muscle_start.append(len(self.muscle_links))
        self.muscle_params.append((f0, lm, lt, lmax, pen))
        self.muscle_activation.append(0.0)
        for l, p in zip(links, positions):
            self.muscle_links.append(l)
            self.muscle_points.append(np.array(p, dtype=np.float64))
        return len(self.muscle_start) - 1

    def add_shape_sphere(self, body, pos=(0.0, 0.0, 0.0), rot=(0.0, 0.0, 0.0, 1.0), radius=1.0, density=1000.0,
                         ke=1.e+5, kd=1000.0, kf=1000.0, mu=0.5):
        self._add_shape(body, pos, rot, GEO_SPHERE, (radius, 0.0, 0.0), density, ke, kd, kf, mu)

    def add_shape_box(self, body, pos=(0.0, 0.0, 0.0), rot=(0.0, 0.0, 0.0, 1.0), hx=0.5, hy=0.5, hz=0.5,
                      density=1000.0, ke=1.e+5, kd=1000.0, kf=1000.0, mu=0.5):
        self._add_shape(body, pos, rot, GEO_BOX, (hx, hy, hz), density, ke, kd, kf, mu)

    def add_shape_capsule(self, body, pos=(0.0, 0.0, 0.0), rot=(0.0, 0.0, 0.0, 1.0), radius=1.0, half_width=0.5,
                          density=1000.0, ke=1.e+5, kd=1000.0, kf=1000.0, mu=0.5):
        self._add_shape(body, pos, rot, GEO_CAPSULE, (radius, half_width, 0.0), density, ke, kd, kf, mu)

    def _add_shape(self, body, pos, rot, kind, scale, density, ke, kd, kf, mu):
        pos = np.array(pos, dtype=np.float64)
        rot = np.array(rot, dtype=np.float64)
        self.shape_body.append(body)
        self.shape_transform.append((pos, rot))
        self.shape_geo_type.append(kind)
        self.shape_geo_scale.append(tuple(float(s) for s in scale))
        self.shape_materials.append((ke, kd, kf, mu))
        m, I = _solid_mass_props(kind, scale, density)
        self._accumulate_mass(body, m, I, pos, rot)

    def _accumulate_mass(self, i, m, I, p, q):
        """Merge a shape's mass into body i: new COM, both inertias shifted to it (model.py:1629-1652)."""
        if i == -1:
            return
        total = self.body_mass[i] + m
        if total == 0.0:
            return
        com = (self.body_com[i] * self.body_mass[i] + p * m) / total
        self.body_inertia[i] = (U.shifted_inertia(self.body_mass[i], self.body_inertia[i], com - self.body_com[i],
                                                  U.quat_identity())
                                + U.shifted_inertia(m, I, com - p, q))
        self.body_mass[i] = total
        self.body_com[i] = com

    # ---- (de)serialisation of the builder state: compiled assets ------------------------------
    _LISTS = ("articulation_start joint_type joint_parent joint_q_start joint_qd_start joint_q joint_qd joint_target "
              "joint_armature joint_limit_lower joint_limit_upper joint_target_ke joint_target_kd joint_limit_ke "
              "joint_limit_kd body_mass shape_body shape_geo_type muscle_start muscle_activation muscle_links").split()

    def save(self, path, **extra):
        """Writes the builder state as a plain .npz ("compiled asset"): lets an environment be built on
        a machine that does not have the original MJCF / URDF / SNU files."""
        d = {k: np.asarray(getattr(self, k)) for k in self._LISTS}
        d["joint_axis"] = np.asarray(self.joint_axis, dtype=np.float64).reshape(-1, 3)
        d["joint_X_pj"] = np.array([U.transform_flatten(x) for x in self.joint_X_pj]).reshape(-1, 7)
        d["body_inertia"] = np.asarray(self.body_inertia, dtype=np.float64).reshape(-1, 3, 3)
        d["body_com"] = np.asarray(self.body_com, dtype=np.float64).reshape(-1, 3)
        d["shape_transform"] = np.array([U.transform_flatten(x) for x in self.shape_transform]).reshape(-1, 7)
        d["shape_geo_scale"] = np.asarray(self.shape_geo_scale, dtype=np.float64).reshape(-1, 3)
        d["shape_materials"] = np.asarray(self.shape_materials, dtype=np.float64).reshape(-1, 4)
        d["muscle_params"] = np.asarray(self.muscle_params, dtype=np.float64).reshape(-1, 5)
        d["muscle_points"] = np.asarray(self.muscle_points, dtype=np.float64).reshape(-1, 3)
        for k, v in extra.items():
            d["extra_" + k] = np.asarray(v)
        np.savez_compressed(path, **d)

    @classmethod
    def load(cls, path):
        b = cls()
        with np.load(path) as d:
            for k in cls._LISTS:
                setattr(b, k, d[k].tolist())
            b.joint_axis = [a for a in d["joint_axis"]]
            b.joint_X_pj = [(x[0:3].copy(), x[3:7].copy()) for x in d["joint_X_pj"]]
            b.body_inertia = [m for m in d["body_inertia"]]
            b.body_com = [c for c in d["body_com"]]
            b.shape_transform = [(x[0:3].copy(), x[3:7].copy()) for x in d["shape_transform"]]
            b.shape_geo_scale = [tuple(s) for s in d["shape_geo_scale"].tolist()]
            b.shape_materials = [tuple(s) for s in d["shape_materials"].tolist()]
            b.muscle_params = [tuple(s) for s in d["muscle_params"].tolist()]
            b.muscle_points = [p for p in d["muscle_points"]]
            b.extras = {k[6:]: d[k] for k in d.files if k.startswith("extra_")}
        return b

    def replicate(self, n):
        """Appends n-1 copies of articulation 0 (what the reference does by re-parsing the asset)."""
        if len(self.articulation_start) != 1:
            raise ValueError("replicate() expects a builder holding exactly one articulation")
        L, nq, nd = len(self.joint_type), len(self.joint_q), len(self.joint_qd)
        ns, nm, nw = len(self.shape_body), len(self.muscle_start), len(self.muscle_links)
        for r in range(1, n):
            self.articulation_start.append(r * L)
            for k in ("joint_type", "joint_axis", "joint_X_pj", "joint_target_ke", "joint_target_kd", "joint_limit_ke",
                      "joint_limit_kd", "body_mass", "body_inertia", "body_com"):
                lst = getattr(self, k)
                lst += lst[:L]
            self.joint_parent += [p + r * L if p >= 0 else -1 for p in self.joint_parent[:L]]
            self.joint_q_start += [s + r * nq for s in self.joint_q_start[:L]]
            self.joint_qd_start += [s + r * nd for s in self.joint_qd_start[:L]]
            for k in ("joint_q", "joint_target", "joint_limit_lower", "joint_limit_upper"):
                lst = getattr(self, k)
                lst += lst[:nq]
            for k in ("joint_qd", "joint_armature"):
                lst = getattr(self, k)
                lst += lst[:nd]
            self.shape_body += [b + r * L for b in self.shape_body[:ns]]
            for k in ("shape_transform", "shape_geo_type", "shape_geo_scale", "shape_materials"):
                lst = getattr(self, k)
                lst += lst[:ns]
            self.muscle_start += [s + r * nw for s in self.muscle_start[:nm]]
            self.muscle_params += self.muscle_params[:nm]
            self.muscle_activation += self.muscle_activation[:nm]
            self.muscle_links += [l + r * L for l in self.muscle_links[:nw]]
            self.muscle_points += self.muscle_points[:nw]

    # ---- template extraction ---------------------------------------------------------------------
    def _template_of(self, a):
        starts = self.articulation_start + [len(self.joint_type)]
        l0, l1 = starts[a], starts[a + 1]
        L = l1 - l0
        q0, q1 = self.joint_q_start[l0], (self.joint_q_start[l1] if l1 < len(self.joint_type) else len(self.joint_q))
        d0, d1 = self.joint_qd_start[l0], (self.joint_qd_start[l1] if l1 < len(self.joint_type) else len(self.joint_qd))
        I_m = np.zeros((L, 6, 6))
        X_cm = np.zeros((L, 7))
        for i in range(L):
            I_m[i, 0:3, 0:3] = self.body_inertia[l0 + i]
            I_m[i, 3, 3] = I_m[i, 4, 4] = I_m[i, 5, 5] = self.body_mass[l0 + i]
            X_cm[i] = U.transform_flatten((self.body_com[l0 + i], U.quat_identity()))
        parent = np.array(self.joint_parent[l0:l1])
        parent = np.where(parent >= 0, parent - l0, -1)
        shapes = [s for s in range(len(self.shape_body)) if l0 <= self.shape_body[s] < l1]
        # muscles whose waypoints live on this articulation
        ms, ml, mp, mpar = [0], [], [], []
        nmus = len(self.muscle_start)
        mstart = self.muscle_start + [len(self.muscle_links)]
        for mi in range(nmus):
            links = self.muscle_links[mstart[mi]:mstart[mi + 1]]
            if links and l0 <= links[0] < l1:
                ml += [l - l0 for l in links]
                mp += self.muscle_points[mstart[mi]:mstart[mi + 1]]
                ms.append(len(ml))
                mpar.append(self.muscle_params[mi])
        t = dict(
            joint_type=self.joint_type[l0:l1], joint_parent=parent,
            joint_q_start=[s - q0 for s in self.joint_q_start[l0:l1]] + [q1 - q0],
            joint_qd_start=[s - d0 for s in self.joint_qd_start[l0:l1]] + [d1 - d0],
            joint_X_pj=np.array([U.transform_flatten(x) for x in self.joint_X_pj[l0:l1]]),
            joint_X_cm=X_cm, joint_axis=np.array(self.joint_axis[l0:l1]), body_I_m=I_m,
            joint_armature=self.joint_armature[d0:d1], joint_target=self.joint_target[q0:q1],
            joint_target_ke=self.joint_target_ke[l0:l1], joint_target_kd=self.joint_target_kd[l0:l1],
            joint_limit_lower=self.joint_limit_lower[q0:q1], joint_limit_upper=self.joint_limit_upper[q0:q1],
            joint_limit_ke=self.joint_limit_ke[l0:l1], joint_limit_kd=self.joint_limit_kd[l0:l1],
            joint_q0=self.joint_q[q0:q1], joint_qd0=self.joint_qd[d0:d1],
            muscle_start=ms, muscle_links=ml, muscle_points=np.array(mp).reshape(-1, 3),
        )
        shape_info = dict(body=[self.shape_body[s] - l0 for s in shapes],
                          transform=[self.shape_transform[s] for s in shapes],
                          geo_type=[self.shape_geo_type[s] for s in shapes],
                          geo_scale=[self.shape_geo_scale[s] for s in shapes],
                          materials=[self.shape_materials[s] for s in shapes])
        return t, shape_info, mpar

    def finalize(self, adapter):
        if not self.articulation_start:
            raise ValueError("no articulation: call add_articulation() before add_link()")
        n_art = len(self.articulation_start)
        t0, shapes0, mpar0 = self._template_of(0)
        # the start pose may differ between replicas (env spacing when rendering); everything else must not
        skip = ("joint_q0", "joint_qd0")
        for a in range(1, n_art):
            ta, sa, _ = self._template_of(a)
            for k in t0:
                if k not in skip and not np.array_equal(np.asarray(t0[k], dtype=np.float64),
                                                        np.asarray(ta[k], dtype=np.float64)):
                    raise NotImplementedError("articulation %d differs from articulation 0 in %s: only replicated "
                                              "environments are supported" % (a, k))
        q0 = np.array(self.joint_q, dtype=np.float32).reshape(n_art, -1)
        qd0 = np.array(self.joint_qd, dtype=np.float32).reshape(n_art, -1)
        return Model(t0, shapes0, mpar0, n_art, q0, qd0, adapter)


class State:
    """Time-varying state (reference: dflex/dflex/model.py:50-130).  Only the tensors that cross the
    DFlexEnv boundary exist; the derived per-substep tensors of the reference live in LDS.
    `joint_act` is allocated (zeros) on first access: the fused env path never touches it."""

    def __init__(self, act_like=None, model=None):
        self.joint_q = None
        self.joint_qd = None
        self._joint_act = None
        self._act_like = act_like
        self._xf_model = model   # whoever produces the state passes the Model (Model.state, the integrator, DFlexEnv.step)
        self._xf_q = None        # joint_q entering the last substep of the step that produced this state, when gradients were on
        self._xf = None

    # Derived tensors of the reference's State (model.py:338-392) that leave LDS only on request: body_X_sc / body_X_sm
    # ([n_envs * n_links, 7], no grad_fn -- the reference's are plain outputs of eval_rigid_fk too).  One small launch on first
    # access (dsim_body_transforms), cached per state object.  After forward() the reference's tensors belong to the joint
    # coordinates that ENTERED the last substep (eval_rigid_fk runs before the integrator, sim.py:2316-2601); the integrator
    # copies them out of the step's checkpoint (`_xf_q`, n_q floats per environment) when gradients are on,
    # so the values match; in no-grad mode there is no checkpoint and the transforms are those of this state's own joint_q
    # (one substep ahead of the reference's: h later).  A state that did not come out of a step (Model.state(), reset) has no
    # such lag in the reference either... it has zeros there; here: the transforms of its joint_q.
    def _body_xf(self):
        if self._xf is None:
            if self._xf_model is None:
                raise RuntimeError("this State was not produced by a Model / integrator: no engine to derive body transforms with")
            eng = self._xf_model.engine()
            q = self._xf_q if self._xf_q is not None else self.joint_q
            self._xf = eng.body_transforms(q)
        return self._xf

    @property
    def body_X_sc(self):
        return self._body_xf()[0]

    @property
    def body_X_sm(self):
        return self._body_xf()[1]

    @property
    def joint_act(self):
        if self._joint_act is None and self._act_like is not None:
            self._joint_act = torch.zeros_like(self._act_like)
        return self._joint_act

    @joint_act.setter
    def joint_act(self, value):
        self._joint_act = value

    def flatten(self):
        return [t for t in (self.joint_q, self.joint_qd, self._joint_act) if torch.is_tensor(t)]


def ground_contacts(shape_info):
    """Static ground-contact points of one articulation (reference: Model.collide, model.py:424-515):
    sphere -> centre, capsule -> the two cap centres, box -> 8 corners."""
    body, point, dist, mat = [], [], [], []
    for s in range(len(shape_info["body"])):
        X = shape_info["transform"][s]
        kind, sc = shape_info["geo_type"][s], shape_info["geo_scale"][s]
        if kind == GEO_SPHERE:
            locs, d = [(0.0, 0.0, 0.0)], np.float32(sc[0])
        elif kind == GEO_CAPSULE:
            hw = float(np.float32(sc[1]))
            locs, d = [(-hw, 0.0, 0.0), (hw, 0.0, 0.0)], np.float32(sc[0])
        elif kind == GEO_BOX:
            e = [float(np.float32(v)) for v in sc]
            locs = [(sx * e[0], sy * e[1], sz * e[2]) for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
            d = np.float32(0.0)
        else:
            continue
        # the reference reads the shape transform back from its float32 tensor (model.py:463)
        X32 = (np.asarray(X[0], np.float32).astype(np.float64), np.asarray(X[1], np.float32).astype(np.float64))
        for p in locs:
            body.append(shape_info["body"][s])
            point.append(U.transform_point(X32, np.array(p)))
            dist.append(float(d))
            mat.append(s)
    return body, point, dist, mat


class Model:
    """N replicas of one articulation.  Attribute names follow dflex/dflex/model.py:136-336 where the
    environments / algorithms touch them (joint_q, joint_qd, muscle_activation, gravity, ground, ...)."""

    def __init__(self, tdict, shape_info, muscle_params, n_art, q0, qd0, adapter):
        self.adapter = adapter
        self.device = torch.device(adapter)
        self._tdict = tdict
        self._shape_info = shape_info
        self.articulation_count = n_art
        self.link_count = len(tdict["joint_type"]) * n_art
        self.links_per_articulation = len(tdict["joint_type"])
        self.joint_coord_count = q0.size
        self.joint_dof_count = qd0.size
        self.coords_per_articulation = q0.shape[1]
        self.dofs_per_articulation = qd0.shape[1]
        self.muscle_count = (len(tdict["muscle_start"]) - 1) * n_art
        self.muscles_per_articulation = len(tdict["muscle_start"]) - 1
        self.muscle_params = np.array(muscle_params, dtype=np.float32).reshape(-1, 5)
        self.shape_count = len(shape_info["body"]) * n_art
        self.particle_count = 0
        self._engine = None
        self._engine_key = None
        self._ground = True
        self._gravity_host = (0.0, -9.8, 0.0)
        self._gravity = torch.tensor(self._gravity_host, dtype=torch.float32, device=self.device)
        self.joint_q = torch.tensor(q0.reshape(-1), dtype=torch.float32, device=self.device)
        self.joint_qd = torch.tensor(qd0.reshape(-1), dtype=torch.float32, device=self.device)
        self.joint_target = torch.tensor(np.tile(np.asarray(tdict["joint_target"], np.float32), n_art),
                                         device=self.device)
        self.muscle_activation = torch.zeros(self.muscle_count, dtype=torch.float32, device=self.device)
        self.contact_count = 0
        self._contacts = None

    # ground / gravity are part of the device-side model: setting them invalidates the engine (no per-step
    # device->host reads are needed to notice a change)
    @property
    def ground(self):
        return self._ground

    @ground.setter
    def ground(self, value):
        self._ground = bool(value)
        self._engine = None

    @property
    def gravity(self):
        return self._gravity

    @gravity.setter
    def gravity(self, value):
        self._gravity_host = tuple(float(x) for x in (value.detach().cpu().tolist() if torch.is_tensor(value) else value))
        self._gravity = torch.tensor(self._gravity_host, dtype=torch.float32, device=self.device)
        self._engine = None

    def state(self):
        s = State(act_like=self.joint_qd, model=self)
        s.joint_q = self.joint_q.clone()
        s.joint_qd = self.joint_qd.clone()
        return s

    def collide(self, state=None):
        """Generates the static ground-contact set (state independent, as in the reference)."""
        self._contacts = ground_contacts(self._shape_info)
        self.contact_count = len(self._contacts[0]) * self.articulation_count
        self._engine = None

    def template(self):
        """The single-articulation template the kernels run on (contacts only if the ground is enabled)."""
        d = dict(self._tdict)
        if self.ground and self._contacts is not None and len(self._contacts[0]):
            body, point, dist, mat = self._contacts
            mats = np.asarray(self._shape_info["materials"], dtype=np.float32).reshape(-1, 4)
            d.update(contact_body=body, contact_point=np.array(point), contact_dist=dist,
                     contact_material=mats[np.asarray(mat, dtype=np.int64)])
        else:
            d.update(contact_body=np.zeros(0, np.int32), contact_point=np.zeros((0, 3)), contact_dist=np.zeros(0),
                     contact_material=np.zeros((0, 4)))
        return ArticulationTemplate(gravity=np.asarray(self._gravity_host, dtype=np.float32), **d)

    def engine(self):
        """Device-side model handle; rebuilt if ground / gravity / contacts changed since the last step."""
        if self._engine is None:
            from ..engine import Engine
            self._engine = Engine(self.template(), self.device)
        return self._engine

    def flatten(self):
        return [v for v in self.__dict__.values() if torch.is_tensor(v)]
