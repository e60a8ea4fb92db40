"""`import envs` -> diffrl_amd.envs (same class names as the reference's envs/__init__.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffrl_amd.envs import *  # noqa: F401,F403,E402
from diffrl_amd.envs import (AntEnv, CartPoleSwingUpEnv, CheetahEnv, DFlexEnv, HopperEnv, HumanoidEnv,  # noqa: F401,E402
                             SNUHumanoidEnv)

from . import _torch_compat  # noqa: E402

_torch_compat.install()   # torch-1.x style indexing of CPU bookkeeping tensors with GPU index tensors (see the module)
