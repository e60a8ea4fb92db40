import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Engine(...)'s default compiles a kernel set for a user model in a background thread and hands it to the NEXT Engine of that model:
# the tests that assert which kernels ran (generic vs specialised) must not depend on what an earlier test left in the cache
os.environ.setdefault("DSIM_AUTO_SPECIALISE", "0")


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


def pytest_sessionstart(session):
    import probe_ledger
    probe_ledger.reset()
    try:
        os.remove(probe_ledger.LEDGER_PATH)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    """Every gradient comparison that passed on a PROBED tolerance instead of the stated 1e-3 (tests/probe_ledger.py) is
    listed at the end of the run, whatever the verbosity: a passing log cannot hide how often the escape hatch was used."""
    import probe_ledger
    tr = terminalreporter
    tr.section("probed tolerances (stated: 1e-3)")
    if not probe_ledger.ENTRIES:
        tr.write_line("none: every gradient comparison of this run passed at the stated 1e-3")
    for tid, (what, whole, cos, n) in probe_ledger.STATED_FORM.items():
        tr.write_line("%s %s: BASELINE.md section 4's own form, no probe: whole-tensor max-norm relative error %.3e (stated 1e-3) -> %s; "
                      "cosine %.6f (stated >= 0.9999) -> %s; %d environments" % (tid, what, whole, "holds" if whole < 1e-3 else "EXCEEDED",
                                                                                 cos, "holds" if cos >= 0.9999 else "BELOW", n))
    for tid, (n, what) in probe_ledger.SAMPLED.items():
        used = sum(1 for r in probe_ledger.ENTRIES if r["test"] == tid)
        tr.write_line("%s %s: %d of %d compared cases probed (%.2f %%)" % (tid, what, used, n, 100.0 * used / max(n, 1)))
    for r in probe_ledger.ENTRIES:
        tr.write_line("%s [%s] %s: error %.2e, reference-order sensitivity %.2e, accepted up to %.2e (case %d of budget %d)"
                      % (r["test"], r["level"], r["what"], r["error"], r["sensitivity"], r["tolerance"], r["used"], r["budget"]))
