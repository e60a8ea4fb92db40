"""Imported by Python's `site` at start-up when dropin/ is on PYTHONPATH (the unmodified-caller runs only).

externals/rl_games/rl_games/algos_torch/sac_agent.py:9 does `from torch.utils.tensorboard import SummaryWriter` at import time,
and torch.utils.tensorboard needs the `tensorboard` package, which this image does not have (no network to install it).  When
-- and only when -- `tensorboard` is missing, a stand-in module with a no-op SummaryWriter is registered under that name, so
that examples/train_rl.py imports unedited.  Logging to TensorBoard is lost; training is not affected."""
import importlib.util
import sys
import types

if importlib.util.find_spec("tensorboard") is None:
    _m = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **kw):
            pass

        def __getattr__(self, name):
            return lambda *a, **kw: None

    _m.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = _m
