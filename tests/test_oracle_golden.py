"""Pins the CPU oracle (oracle/dsim_oracle.cpp) to golden vectors generated from the real reference
(oracle/gen_golden.py).  CPU-only."""
import numpy as np
import pytest

from oracle_lib import golden, oracle_backward, oracle_forward, template_from_golden

ENVS = ["cartpole", "ant", "humanoid", "snu", "hopper", "cheetah"]


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("env", ENVS)
def test_first_substep_intermediates(env):
    t = template_from_golden(env)
    g = golden(env + "_step")
    h = float(g["sub_dt"])
    mact = g.get("muscle_act_in")
    q1, qd1, dbg = oracle_forward(t, g["q_in"], g["qd_in"], g["act_in"], mact, h, 1, 1, debug=True)
    # the restatement follows the reference's operation order -> (almost) bit-exact
    for name in ["X_sc", "X_sm", "S_s", "I_s", "v_s", "a_s", "f_s", "ft_s", "tau", "H", "L", "qdd"]:
        assert relerr(dbg[name], g["sub_" + name]) < 1e-6, name
    assert relerr(q1, g["sub_q"]) < 1e-6
    assert relerr(qd1, g["sub_qd"]) < 1e-6


@pytest.mark.parametrize("env", ENVS)
def test_env_step_forward(env):
    t = template_from_golden(env)
    g = golden(env + "_step")
    qo, qdo, _ = oracle_forward(t, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), float(g["dt"]),
                                int(g["substeps"]), int(g["mm_freq"]))
    assert relerr(qo, g["q_out"]) < 1e-5
    assert relerr(qdo, g["qd_out"]) < 1e-5


@pytest.mark.parametrize("env", ENVS)
def test_env_step_adjoint(env):
    t = template_from_golden(env)
    g = golden(env + "_step")
    r = oracle_backward(t, g["q_in"], g["qd_in"], g["act_in"], g.get("muscle_act_in"), float(g["dt"]),
                        int(g["substeps"]), int(g["mm_freq"]), g["gq_out"], g["gqd_out"])
    assert relerr(r["q_out"], g["q_out"]) < 1e-5
    # stated fp32 tolerance for one env-step of gradients: 1e-4 relative (max-norm), BASELINE.md section 4
    # (measured: <= 1.4e-5; the oracle's adjoint is taped AD, the reference's is generated code)
    assert relerr(r["gq"], g["gq_in"]) < 5e-5
    assert relerr(r["gqd"], g["gqd_in"]) < 5e-5
    if "gact_in" in g:
        assert relerr(r["gact"], g["gact_in"]) < 5e-5
    if "gmuscle_act_in" in g:
        assert relerr(r["gmact"], g["gmuscle_act_in"]) < 5e-5
