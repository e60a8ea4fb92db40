import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the driver's GPU tier)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a real MI355X: on a box without one they are skipped, not failed (the product itself still
    fails loudly there, see tests/test_capi_cpu.py::test_engine_refuses_cpu)."""
    import pytest
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="needs a GPU (run through gpurun / the driver's GPU tier)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
