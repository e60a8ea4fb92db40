"""gym.spaces stand-in: Box is the class diffrl_amd's environments already expose as observation_space / action_space."""
import os
import sys

try:
    from diffrl_amd.envs.dflex_env import Box  # noqa: F401
except ImportError:   # the repository root is not on sys.path yet (PYTHONPATH=<repo>/dropin only): add it, once
    sys.path.append(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from diffrl_amd.envs.dflex_env import Box  # noqa: F401,E402


class Space:
    pass


class Discrete(Space):
    def __init__(self, n):
        self.n = n


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = nvec


class Tuple(Space, tuple):
    pass


class Dict(Space, dict):
    pass
