"""The stand-in itself (see __init__.py for when it is used)."""


def _unavailable(*a, **kw):
    raise RuntimeError("ray stand-in (dropin/ray): Ray workers are not available; the DFLEX vecenv does not use them")


init = get = remote = _unavailable
