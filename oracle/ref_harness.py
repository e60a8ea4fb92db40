"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (diffrl_amd/).

Loads the *real* reference (NVlabs/DiffRL, mounted read-only at /root/reference)
in THIS container so that golden vectors can be generated from it
(oracle/gen_golden.py) and the C restatement (oracle/dsim_oracle.cpp) can be
pinned against it.  `/root/reference` does not exist on the GPU box, therefore
nothing under tests/ -m gpu, smoke() or bench.py may call this module.

What it does (all outside the arithmetic of the reference):
  * copies /root/reference/dflex/dflex to a scratch dir (the package JIT-writes
    `kernels/` next to itself, dflex/dflex/adjoint.py:1813-1818) and patches the
    ast.Subscript handler for Python >= 3.9 (adjoint.py:1103-1119 uses
    node.slice.value, which no longer exists),
  * stubs `gym`, `urdfpy`, `tensorboardX`, restores `numpy.Inf`,
  * imports the reference `envs` package.
"""
import importlib
import os
import shutil
import sys
import types
import xml.etree.ElementTree as ET

import numpy as np

REF_ROOT = os.environ.get("DIFFRL_REFERENCE", "/root/reference")
SCRATCH = os.environ.get("DSIM_REF_SCRATCH", "/tmp/dsim_ref_scratch")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "dflex", "dflex"))


def _patch_adjoint(path):
    src = open(path).read()
    if "_dsim_sl" in src:
        return
    old_a = "if isinstance(node.slice.value, ast.Tuple):"
    assert old_a in src
    src = src.replace(
        old_a,
        "_dsim_sl = node.slice.value if isinstance(node.slice, getattr(ast, 'Index', ())) else node.slice\n"
        "                if isinstance(_dsim_sl, ast.Tuple):")
    src = src.replace("for arg in node.slice.value.elts:", "for arg in _dsim_sl.elts:")
    src = src.replace("var = adj.eval(node.slice.value)", "var = adj.eval(_dsim_sl)")
    open(path, "w").write(src)


# ---- urdfpy stand-in: just enough of URDF.load for cartpole.urdf -------------
def _xyz_rpy_matrix(xyz, rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    M = np.eye(4)
    M[:3, :3] = R
    M[:3, 3] = xyz
    return M


def _matrix_to_xyz_rpy(M):
    xyz = M[:3, 3]
    R = M[:3, :3]
    p = -np.arcsin(np.clip(R[2, 0], -1, 1))
    r = np.arctan2(R[2, 1], R[2, 2])
    y = np.arctan2(R[1, 0], R[0, 0])
    return np.array([xyz[0], xyz[1], xyz[2], r, p, y])


def _origin_of(elem):
    o = elem.find("origin") if elem is not None else None
    xyz = [0.0, 0.0, 0.0]
    rpy = [0.0, 0.0, 0.0]
    if o is not None:
        if o.get("xyz"):
            xyz = [float(v) for v in o.get("xyz").split()]
        if o.get("rpy"):
            rpy = [float(v) for v in o.get("rpy").split()]
    return _xyz_rpy_matrix(xyz, rpy)


class _NS(types.SimpleNamespace):
    pass


def _make_urdfpy():
    mod = types.ModuleType("urdfpy")

    class URDF:
        @staticmethod
        def load(path):
            root = ET.parse(path).getroot()
            links, link_map = [], {}
            for l in root.findall("link"):
                cols = []
                for c in l.findall("collision"):
                    g = c.find("geometry")
                    geo = _NS(box=None, sphere=None, cylinder=None, mesh=None)
                    if g.find("box") is not None:
                        geo.box = _NS(size=np.array([float(v) for v in g.find("box").get("size").split()]))
                    if g.find("sphere") is not None:
                        geo.sphere = _NS(radius=float(g.find("sphere").get("radius")))
                    if g.find("cylinder") is not None:
                        geo.cylinder = _NS(radius=float(g.find("cylinder").get("radius")),
                                           length=float(g.find("cylinder").get("length")))
                    cols.append(_NS(origin=_origin_of(c), geometry=geo))
                link = _NS(name=l.get("name"), collisions=cols)
                links.append(link)
                link_map[link.name] = link
            joints = []
            for j in root.findall("joint"):
                ax = j.find("axis")
                axis = np.array([float(v) for v in ax.get("xyz").split()]) if ax is not None else np.array([1.0, 0, 0])
                lim = j.find("limit")
                limit = None
                if lim is not None:
                    limit = _NS(lower=float(lim.get("lower")) if lim.get("lower") is not None else None,
                                upper=float(lim.get("upper")) if lim.get("upper") is not None else None)
                dyn = j.find("dynamics")
                dynamics = None
                if dyn is not None:
                    dynamics = _NS(damping=float(dyn.get("damping", 0.0)))
                joints.append(_NS(name=j.get("name"), joint_type=j.get("type"), axis=axis,
                                  parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                                  origin=_origin_of(j), limit=limit, dynamics=dynamics))
            return _NS(links=links, joints=joints, link_map=link_map)

    mod.URDF = URDF
    mod.matrix_to_xyz_rpy = _matrix_to_xyz_rpy
    return mod


def _install_stubs():
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")
        spaces = types.ModuleType("gym.spaces")

        class Box:
            def __init__(self, low, high, shape=None, dtype=None):
                self.low, self.high = low, high
                self.shape = np.shape(low)

        spaces.Box = Box
        gym.spaces = spaces
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces
    if "urdfpy" not in sys.modules:
        sys.modules["urdfpy"] = _make_urdfpy()
    if "tensorboardX" not in sys.modules:
        tb = types.ModuleType("tensorboardX")

        class SummaryWriter:
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

            def flush(self):
                pass

            def close(self):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["tensorboardX"] = tb


_loaded = {}


def load_reference():
    """Returns (dflex_module, envs_module) of the real reference, CPU path."""
    if _loaded:
        return _loaded["df"], _loaded["envs"]
    if not reference_available():
        raise RuntimeError("reference checkout not found at %s" % REF_ROOT)
    dst = os.path.join(SCRATCH, "dflex")
    if not os.path.isdir(dst):
        os.makedirs(SCRATCH, exist_ok=True)
        shutil.copytree(os.path.join(REF_ROOT, "dflex", "dflex"), dst,
                        ignore=shutil.ignore_patterns("kernels", "__pycache__"))
    _patch_adjoint(os.path.join(dst, "adjoint.py"))
    _install_stubs()
    # the scratch copy must win over both the namespace package /root/reference/dflex
    # and this repo's own drop-in `dflex`
    for m in [k for k in sys.modules if k == "dflex" or k.startswith("dflex.") or k == "envs" or k.startswith("envs.")]:
        del sys.modules[m]
    sys.path.insert(0, SCRATCH)
    df = importlib.import_module("dflex")
    sys.path.insert(1, REF_ROOT)
    envs = importlib.import_module("envs")
    # envs/dflex_env.py does sys.path.insert(0, project_root); keep scratch first
    if sys.path[0] != SCRATCH:
        sys.path.remove(SCRATCH)
        sys.path.insert(0, SCRATCH)
    _loaded["df"], _loaded["envs"] = df, envs
    return df, envs


if __name__ == "__main__":
    df, envs = load_reference()
    print("reference loaded:", df.__file__, envs.__file__)
