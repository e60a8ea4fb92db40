"""Minimal stand-in used only when the real tensorboardX is absent (algorithms/shac.py:20 imports it)."""


class SummaryWriter:
    def __init__(self, *args, **kwargs):
        pass

    def add_scalar(self, *args, **kwargs):
        pass

    def flush(self):
        pass

    def close(self):
        pass
