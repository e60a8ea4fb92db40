"""Drop-in `dflex` namespace for the articulated rigid-body path (what envs/ and algorithms/ of
NVlabs/DiffRL import): `df.sim.ModelBuilder`, `df.sim.SemiImplicitIntegrator`, `df.config`, the
joint-type constants and the build-time transform helpers.  No JIT, no code generation: importing
this module only loads the prebuilt HIP library on first use."""
from . import config, sim, util  # noqa: F401
from .model import (GEO_BOX, GEO_CAPSULE, GEO_SPHERE, Model, ModelBuilder, State)  # noqa: F401
from .sim import SemiImplicitIntegrator  # noqa: F401
from .util import (normalize, quat_from_axis_angle, quat_identity, quat_inverse, quat_multiply, quat_rotate,  # noqa: F401
                   quat_to_matrix, rpy2quat, transform, transform_identity, transform_multiply, transform_point)
from ..template import JOINT_BALL, JOINT_FIXED, JOINT_FREE, JOINT_PRISMATIC, JOINT_REVOLUTE  # noqa: F401
