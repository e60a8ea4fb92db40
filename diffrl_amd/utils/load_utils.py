"""Asset loaders that fill a `ModelBuilder`: MJCF (ant / humanoid / hopper / half-cheetah), the SNU
skeleton + muscle XML, and a minimal URDF reader (cartpole) that needs no third-party package.

Semantics follow the reference loaders (utils/load_utils.py: `urdf_load` 129-229, `parse_mjcf`
289-488, `Skeleton` 502-718) closely enough that the resulting model constants are identical; the
tests pin that against `tests/golden/<env>_model.npz`.  Notable inherited behaviour:
  * MJCF: one link per <joint> of a body (extra joints become mass-less intermediate links), geoms go
    to the last of them; <default> classes and <actuator> are ignored; only sphere / capsule geoms;
  * URDF: masses come from the collision boxes and the density, <inertial> is ignored.
"""
import math
import os
import xml.etree.ElementTree as ET

import numpy as np

from .. import dflex as df

_MJCF_JOINT = {"ball": df.JOINT_BALL, "hinge": df.JOINT_REVOLUTE, "slide": df.JOINT_PRISMATIC,
               "free": df.JOINT_FREE, "fixed": df.JOINT_FIXED}


def _vec(node, key, default):
    return np.array([float(v) for v in node.attrib[key].split()]) if key in node.attrib else np.array(default, dtype=np.float64)


def set_seed(seed, torch_deterministic=False):
    """Same contract as the reference helper (utils/load_utils.py:25-49), minus the CUDA-only knobs."""
    import random

    import torch
    if seed == -1:
        seed = 42 if torch_deterministic else int(np.random.randint(0, 10000))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    return seed


# --------------------------------------------------------------------------------------------------
# MJCF
def parse_mjcf(filename, builder, density=1000.0, stiffness=0.0, damping=1.0, contact_ke=1e4, contact_kd=1e4,
               contact_kf=1e3, contact_mu=0.5, limit_ke=100.0, limit_kd=10.0, armature=0.01, radians=False,
               load_stiffness=False, load_armature=False):
    root = ET.parse(filename).getroot()
    mat = dict(ke=contact_ke, kd=contact_kd, kf=contact_kf, mu=contact_mu)

    def joint_range(j):
        limited = (j.attrib["limited"] == "true") if "limited" in j.attrib else True
        if not limited:
            return np.array([-1.e+6, 1.e+6])
        if radians:
            return _vec(j, "range", (np.deg2rad(-170.0), np.deg2rad(170.0)))
        return np.deg2rad(_vec(j, "range", (-170.0, 170.0)))

    def add_geom(link, geom, anchor):
        kind = geom.attrib["type"]
        size = _vec(geom, "size", [1.0])
        pos = _vec(geom, "pos", (0.0, 0.0, 0.0))
        rot = _vec(geom, "quat", (0.0, 0.0, 0.0, 1.0))
        if kind == "sphere":
            builder.add_shape_sphere(link, pos=pos - anchor, rot=rot, radius=size[0], density=density, **mat)
        elif kind == "capsule":
            if "fromto" in geom.attrib:
                ft = _vec(geom, "fromto", (0.0, 0.0, 0.0, 1.0, 0.0, 0.0))
                a, b = ft[0:3], ft[3:6]
                direction = df.normalize(b - a)
                # rotate the builder's x-aligned capsule onto the fromto direction
                angle = math.acos(np.dot(direction, (1.0, 0.0, 0.0)))
                rot_axis = df.normalize(np.cross(direction, (1.0, 0.0, 0.0)))
                pos = (a + b) * 0.5
                rot = df.quat_from_axis_angle(rot_axis, -angle)
                radius, half = size[0], np.linalg.norm(b - a) * 0.5
            else:
                radius, half = size[0], size[1]
                if "axisangle" in geom.attrib:
                    aa = _vec(geom, "axisangle", (0.0, 1.0, 0.0, 0.0))
                    rot = df.quat_from_axis_angle(aa[0:3], aa[3])
                if "quat" in geom.attrib:
                    rot = _vec(geom, "quat", df.quat_identity())
                rot = df.quat_multiply(rot, df.quat_from_axis_angle((0.0, 1.0, 0.0), -math.pi * 0.5))
            builder.add_shape_capsule(link, pos=pos - anchor, rot=rot, radius=radius, half_width=half,
                                      density=density, **mat)
        else:
            print("MJCF geom type '%s' is not supported, skipped" % kind)

    def walk(body, parent, anchor):
        offset = _vec(body, "pos", (0.0, 0.0, 0.0))
        link = parent
        for j in body.findall("joint"):
            jpos = _vec(j, "pos", (0.0, 0.0, 0.0))
            rng = joint_range(j)
            if parent == -1:
                offset = np.zeros(3)  # the root body is placed by the environment, not by the asset
            link = builder.add_link(
                parent, X_pj=df.transform(offset + jpos - anchor, df.quat_identity()),
                axis=df.normalize(_vec(j, "axis", (0.0, 0.0, 0.0))), type=_MJCF_JOINT[j.attrib.get("type", "hinge")],
                limit_lower=rng[0], limit_upper=rng[1], limit_ke=limit_ke, limit_kd=limit_kd,
                stiffness=float(j.attrib["stiffness"]) if (load_stiffness and "stiffness" in j.attrib) else stiffness,
                damping=float(j.attrib["damping"]) if "damping" in j.attrib else damping,
                armature=float(j.attrib["armature"]) if (load_armature and "armature" in j.attrib) else armature)
            parent, offset, anchor = link, np.zeros(3), jpos
        for geom in body.findall("geom"):
            add_geom(link, geom, anchor)
        for child in body.findall("body"):
            walk(child, link, anchor)

    builder.add_articulation()
    for body in root.find("worldbody").findall("body"):
        walk(body, -1, np.zeros(3))


# --------------------------------------------------------------------------------------------------
# URDF (subset: links with <collision> boxes / spheres / cylinders, prismatic / revolute / continuous /
# fixed / floating joints) -- replaces the reference's dependency on the `urdfpy` package
def _rpy_matrix(r, p, y):
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _origin_xyz_rpy(elem):
    """xyz + roll/pitch/yaw of an <origin>, after the matrix round trip urdfpy performs."""
    o = elem.find("origin")
    xyz = [float(v) for v in o.attrib.get("xyz", "0 0 0").split()] if o is not None else [0.0, 0.0, 0.0]
    rpy = [float(v) for v in o.attrib.get("rpy", "0 0 0").split()] if o is not None else [0.0, 0.0, 0.0]
    R = _rpy_matrix(*rpy)
    pitch = -math.asin(min(1.0, max(-1.0, R[2, 0])))
    return np.array(xyz), (math.atan2(R[2, 1], R[2, 2]), pitch, math.atan2(R[1, 0], R[0, 0]))


def urdf_load(builder, filename, xform, floating=False, armature=0.0, shape_ke=1.e+4, shape_kd=1.e+4, shape_kf=1.e+2,
              shape_mu=0.25, limit_ke=100.0, limit_kd=1.0):
    robot = ET.parse(filename).getroot()
    links = {l.attrib["name"]: l for l in robot.findall("link")}
    first = robot.findall("link")[0].attrib["name"]
    mat = dict(ke=shape_ke, kd=shape_kd, kf=shape_kf, mu=shape_mu)

    def add_collisions(index, link_elem):
        for col in link_elem.findall("collision"):
            pos, rpy = _origin_xyz_rpy(col)
            rot = df.rpy2quat(*rpy)
            geo = col.find("geometry")
            box, sph, cyl = geo.find("box"), geo.find("sphere"), geo.find("cylinder")
            if box is not None:
                sx, sy, sz = [float(v) for v in box.attrib["size"].split()]
                builder.add_shape_box(index, pos, rot, sx * 0.5, sy * 0.5, sz * 0.5, **mat)
            if sph is not None:
                builder.add_shape_sphere(index, pos, rot, float(sph.attrib["radius"]), **mat)
            if cyl is not None:  # URDF cylinders are z-aligned, the builder's capsules x-aligned
                turn = df.quat_from_axis_angle((0.0, 1.0, 0.0), math.pi * 0.5)
                builder.add_shape_capsule(index, pos, df.quat_multiply(rot, turn), float(cyl.attrib["radius"]),
                                          float(cyl.attrib["length"]) * 0.5, **mat)

    builder.add_articulation()
    if floating:
        base = builder.add_link(-1, df.transform_identity(), (0, 0, 0), df.JOINT_FREE)
        s = builder.joint_q_start[base]
        builder.joint_q[s:s + 3] = [float(v) for v in xform[0]]
        builder.joint_q[s + 3:s + 7] = [float(v) for v in xform[1]]
    else:
        base = builder.add_link(-1, xform, (0, 0, 0), df.JOINT_FIXED)
    add_collisions(base, links[first])
    index = {first: base}
    kinds = {"revolute": df.JOINT_REVOLUTE, "continuous": df.JOINT_REVOLUTE, "prismatic": df.JOINT_PRISMATIC,
             "fixed": df.JOINT_FIXED, "floating": df.JOINT_FREE}
    for j in robot.findall("joint"):
        kind = kinds[j.attrib["type"]]
        axis = (0.0, 0.0, 0.0)
        if kind in (df.JOINT_REVOLUTE, df.JOINT_PRISMATIC):
            ax = j.find("axis")
            axis = [float(v) for v in ax.attrib["xyz"].split()] if ax is not None else [1.0, 0.0, 0.0]
        pos, rpy = _origin_xyz_rpy(j)
        lower, upper = -1.e+3, 1.e+3
        lim = j.find("limit")
        if lim is not None:
            lower = float(lim.attrib["lower"]) if "lower" in lim.attrib else lower
            upper = float(lim.attrib["upper"]) if "upper" in lim.attrib else upper
        dyn = j.find("dynamics")
        damping = float(dyn.attrib["damping"]) if (dyn is not None and float(dyn.attrib.get("damping", 0.0))) else 0.0
        parent = index.get(j.find("parent").attrib["link"], -1)
        child = j.find("child").attrib["link"]
        link = builder.add_link(parent=parent, X_pj=df.transform(pos, df.rpy2quat(*rpy)), axis=axis, type=kind,
                                limit_lower=lower, limit_upper=upper, limit_ke=limit_ke, limit_kd=limit_kd,
                                damping=damping)
        add_collisions(link, links[child])
        index[child] = link


# --------------------------------------------------------------------------------------------------
# SNU musculoskeletal model (skeleton XML + muscle XML)
class MuscleUnit:
    def __init__(self, name, strength):
        self.name = name
        self.bones = []
        self.points = []
        self.muscle_strength = strength


class Skeleton:
    _JOINT = {"Ball": df.JOINT_BALL, "Revolute": df.JOINT_REVOLUTE, "Prismatic": df.JOINT_PRISMATIC,
              "Free": df.JOINT_FREE, "Fixed": df.JOINT_FIXED}

    def __init__(self, skeleton_file, muscle_file, builder, filter={}, visualize_shapes=True, stiffness=5.0,
                 damping=2.0, contact_ke=5000.0, contact_kd=2000.0, contact_kf=1000.0, contact_mu=0.5,
                 limit_ke=1000.0, limit_kd=10.0, armature=0.05):
        self.armature, self.stiffness, self.damping = armature, stiffness, damping
        self.contact = dict(ke=contact_ke, kd=contact_kd, kf=contact_kf, mu=contact_mu)
        self.limit_ke, self.limit_kd = limit_ke, limit_kd
        self.node_map, self.xform_map, self.mesh_map = {}, {}, {}
        self.muscles = []
        self._read_skeleton(skeleton_file, builder, filter)
        if muscle_file is not None:
            self._read_muscles(muscle_file, builder)

    @staticmethod
    def _frame(elem):
        t = elem.find("Transformation")
        R = np.array([float(v) for v in t.attrib["linear"].split()]).reshape(3, 3)
        p = np.array([float(v) for v in t.attrib["translation"].split()])
        return df.transform(p, df.util.quat_from_matrix(R))

    def _read_skeleton(self, filename, builder, keep):
        self.coord_start = len(builder.joint_q)
        self.dof_start = len(builder.joint_qd)
        builder.add_articulation()
        for node in ET.parse(filename).getroot():
            if node.tag != "Node":
                continue
            name, parent_name = node.attrib["name"], node.attrib["parent"]
            body, joint = node.find("Body"), node.find("Joint")
            parent_link, parent_X = -1, df.transform_identity()
            if parent_name in self.node_map:
                parent_link, parent_X = self.node_map[parent_name], self.xform_map[parent_name]
            size = np.array([float(v) for v in body.attrib["size"].split()])
            mass = float(body.attrib["mass"])
            rel_mass = mass / 15.0  # gains are scaled relative to a 15 kg segment
            body_X, joint_X = self._frame(body), self._frame(joint)
            kind = self._JOINT[joint.attrib["type"]]
            lower, upper = -1.e+3, 1.e+3
            if kind == df.JOINT_REVOLUTE:
                if "lower" in joint.attrib:
                    lower = float(joint.attrib["lower"].split()[0])
                if "upper" in joint.attrib:
                    upper = float(joint.attrib["upper"].split()[0])
            axis = _vec(joint, "axis", (0.0, 0.0, 0.0))
            link = -1
            if len(keep) == 0 or name in keep:
                X_pj = df.transform_multiply(df.util.transform_inverse(parent_X), joint_X)
                X_body = df.transform_multiply(df.util.transform_inverse(joint_X), body_X)
                if parent_link == -1:
                    X_pj = df.transform_identity()
                link = builder.add_link(parent=parent_link, X_pj=X_pj, axis=axis, type=kind, limit_lower=lower,
                                        limit_upper=upper, limit_ke=self.limit_ke * rel_mass,
                                        limit_kd=self.limit_kd * rel_mass, damping=self.damping,
                                        stiffness=self.stiffness * math.sqrt(rel_mass), armature=self.armature)
                builder.add_shape_box(body=link, pos=X_body[0], rot=X_body[1], hx=size[0] * 0.5, hy=size[1] * 0.5,
                                      hz=size[2] * 0.5, density=mass / (size[0] * size[1] * size[2]), **self.contact)
            self.xform_map[name] = joint_X
            self.node_map[name] = link
            self.mesh_map[os.path.splitext(body.attrib["obj"])[0]] = link

    def _read_muscles(self, filename, builder):
        self.muscle_start = len(builder.muscle_activation)
        for unit in ET.parse(filename).getroot():
            if unit.tag != "Unit":
                continue
            mu = MuscleUnit(unit.attrib["name"], float(unit.attrib["f0"]))
            complete = True
            for wp in unit.iter("Waypoint"):
                bone = wp.attrib["body"]
                link = self.node_map[bone]
                if link == -1:
                    complete = False
                    break
                local = np.array([float(v) for v in wp.attrib["p"].split()], dtype=np.float32)
                mu.bones.append(link)
                mu.points.append(df.transform_point(df.util.transform_inverse(self.xform_map[bone]), local))
            if complete:
                self.muscles.append(mu)
                builder.add_muscle(mu.bones, mu.points, f0=float(unit.attrib["f0"]), lm=float(unit.attrib["lm"]),
                                   lt=float(unit.attrib["lt"]), lmax=float(unit.attrib["lmax"]),
                                   pen=float(unit.attrib["pen_angle"]))
