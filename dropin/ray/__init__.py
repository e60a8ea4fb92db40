"""Stand-in for `ray`: externals/rl_games/rl_games/common/vecenv.py imports it at module level for its RayVecEnv, which the
DFlexEnv path (vecenv type 'DFLEX', examples/train_rl.py:81-84) never instantiates."""
import importlib.machinery as _mach
import importlib.util as _util
import os as _os
import sys as _sys

_dropin = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_real = _mach.PathFinder.find_spec(__name__, [p for p in _sys.path if _os.path.abspath(p or ".") != _dropin])
if _real is not None and _real.loader is not None:
    # the real package is installed somewhere behind this directory on sys.path: step aside and load it under this name
    _m = _util.module_from_spec(_real)
    _sys.modules[__name__] = _m
    _real.loader.exec_module(_m)
else:
    from ._standin import *  # noqa: F401,F403
