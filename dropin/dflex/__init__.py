"""`import dflex as df` -> diffrl_amd.dflex (no JIT, no code generation)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from diffrl_amd.dflex import *  # noqa: F401,F403,E402
from diffrl_amd.dflex import config, sim, util  # noqa: F401,E402
