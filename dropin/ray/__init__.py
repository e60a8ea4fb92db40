"""Stand-in for `ray`: externals/rl_games/rl_games/common/vecenv.py imports it at module level for its RayVecEnv, which the
DFlexEnv path (vecenv type 'DFLEX', examples/train_rl.py:81-84) never instantiates."""


def _unavailable(*a, **kw):
    raise RuntimeError("ray stand-in (dropin/ray): Ray workers are not available; the DFLEX vecenv does not use them")


init = get = remote = _unavailable
