"""The probed-tolerance escape hatch of the gradient parity tests (tests/probe_ledger.py) is bounded, counted and loud --
and it cannot absorb a real adjoint defect: with a deliberate 1 % error injected into ONE adjoint phase (developer flag
-DDSIM_INJECT_ADJ_ERROR=1.01f, dsim_core.hpp: dsim_bwd_bodies) the parity tests fail, on the host harness and on the GPU."""
import os
import subprocess
import sys

import pytest

import probe_ledger

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INJECT_LIB = os.path.join(ROOT, "tests", "inject", "libdsim_inject.so")


def test_ledger_caps_counts_and_records(tmp_path, monkeypatch):
    monkeypatch.setattr(probe_ledger, "LEDGER_PATH", str(tmp_path / "ledger.jsonl"))
    monkeypatch.setenv("PYTEST_CURRENT_TEST", "tests/x.py::case (call)")
    n0 = len(probe_ledger.ENTRIES)
    try:
        with pytest.warns(UserWarning, match="PROBED TOLERANCE"):
            assert probe_ledger.accept("step", 2e-3, 1e-3, budget=2) == pytest.approx(3e-3)      # 3 x sensitivity
        with pytest.warns(UserWarning):
            assert probe_ledger.accept("step", 4e-2, 1.0, budget=2) == 5e-3                       # hard ceiling of the level
        with pytest.warns(UserWarning):
            assert probe_ledger.accept("rollout", 2e-3, 1e-6, budget=3) == 1e-3                   # never below the stated 1e-3
        with pytest.warns(UserWarning), pytest.raises(AssertionError, match="budget"):
            probe_ledger.accept("rollout", 2e-3, 1.0, budget=3)                                   # the 4th case of a budget of 3
        assert len(open(probe_ledger.LEDGER_PATH).read().splitlines()) == 4
    finally:
        del probe_ledger.ENTRIES[n0:]               # these four are not findings of this run
        probe_ledger._USED.pop("tests/x.py::case", None)


def _run_parity(test_id, env):
    e = dict(os.environ)
    e.update(env)
    e.pop("PYTEST_CURRENT_TEST", None)
    e["DSIM_PROBE_LEDGER"] = os.path.join(ROOT, "gpurun_out", "probe_ledger_injected.jsonl")   # not the outer run's ledger
    r = subprocess.run([sys.executable, "-m", "pytest", test_id, "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    return r.returncode, r.stdout


def test_injected_adjoint_error_fails_the_host_harness_parity_tests():
    """Ant H = 32 rollout and the humanoid episode recording (which has a probe budget of one case) against the harness built
    with the injected error: both must fail; against the regular harness they pass (that is the rest of the suite)."""
    for tid in ("tests/test_reference_episodes.py::test_emu_h32_rollout_vs_reference[ant]",
                "tests/test_reference_episodes.py::test_emu_episode_rollout_vs_reference[humanoid]"):
        rc, out = _run_parity(tid, {"DSIM_EMU_LIB": "libdsim_emu_inject.so"})
        assert rc != 0 and "1 failed" in out, out[-2000:]
        assert "AssertionError" in out or "assert" in out


@pytest.mark.gpu
def test_injected_adjoint_error_fails_the_fullsize_humanoid_test():
    """BASELINE configs[2] (Humanoid 1024 x 32) against a library built with the injected error: every sampled environment
    lands above 1e-3, which exceeds the budget of branch-boundary environments at once -- no probing, no pass."""
    if not os.path.exists(INJECT_LIB):
        pytest.fail("tests/inject/libdsim_inject.so is missing: __graft_entry__.build() compiles it")
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "humanoid_1024x32.npz")):
        pytest.skip("recording not generated")
    rc, out = _run_parity("tests/test_reference_episodes.py::test_gpu_fullsize_humanoids_vs_reference[humanoid_1024x32]",
                          {"DSIM_LIB": INJECT_LIB})
    assert rc != 0 and "1 failed" in out, out[-2000:]
    assert "sampled environments above 1e-3" in out, out[-2000:]
