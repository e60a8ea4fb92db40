"""Stand-in for the `gym` package, only as far as the reference's callers import it (externals/rl_games, envs/dflex_env.py:18):
base classes that rl_games' wrappers subclass and the `spaces` it inspects.  Used only when the real package is absent: __init__ looks for an installed `gym` behind this directory on sys.path first
(this directory sits on PYTHONPATH for the unmodified-caller runs; the diffrl_amd package itself never imports gym)."""
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
