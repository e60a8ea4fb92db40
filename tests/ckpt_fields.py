"""Test helper: the forward intermediates of the FIRST substep, read back from a checkpoint (full mode: row 0 of an
environment's checkpoint is the saved block of substep 0 -- q, qd, X_sc, S, v, a, 10-parameter world inertias, f_tot, qdd --
at the offsets of dsim_layout.hpp), in the shapes of the reference's recordings (tests/golden/<env>_step.npz: sub_*)."""
import numpy as np

from emu_lib import layout


def dense_inertia(p):
    """10 parameters (m, h = m c, A about the world origin) -> the reference's dense 6 x 6 spatial inertia (w, v ordering)"""
    m, hx, hy, hz, axx, axy, axz, ayy, ayz, azz = [float(x) for x in p]
    A = np.array([[axx, axy, axz], [axy, ayy, ayz], [axz, ayz, azz]])
    H = np.array([[0.0, -hz, hy], [hz, 0.0, -hx], [-hy, hx, 0.0]])
    out = np.zeros((6, 6))
    out[:3, :3], out[:3, 3:], out[3:, :3], out[3:, 3:] = A, H, H.T, m * np.eye(3)
    return out


def first_substep(t, ckpt):
    off, dims = layout(t)
    L, nd = t.n_links, t.n_qd
    base = off["q"]
    ck = np.asarray(ckpt, np.float32)

    def field(name, n, shape):
        a = ck[:, off[name] - base:off[name] - base + n]
        return a.reshape((ck.shape[0],) + shape)

    out = dict(X_sc=field("xsc", 7 * L, (L, 7)), S_s=field("S", 6 * nd, (nd, 6)), v_s=field("v", 6 * L, (L, 6)),
               a_s=field("a", 6 * L, (L, 6)), ft_s=field("ftot", 6 * L, (L, 6)), qdd=field("qdd", nd, (nd,)))
    i10 = field("i10", 10 * L, (L, 10))
    out["I_s"] = np.stack([[dense_inertia(p) for p in env] for env in i10]).astype(np.float32)
    return out


def compare_with_reference(t, fields, g, relerr):
    """max-norm relative errors of the first-substep intermediates against the reference's recording g (sub_* arrays).
    The motion subspace of a free joint is the identity and is not stored in the checkpoint: only the other dofs are compared;
    the reference's ft_s holds the descendants' part of f_tot only (sim.py:1792-1842), f_tot = f_s + ft_s."""
    jt, qds = np.asarray(t.joint_type), np.asarray(t.joint_qd_start)
    dofs = [d for i in range(t.n_links) if jt[i] != 4 for d in range(qds[i], qds[i + 1])]
    err = {k: relerr(fields[k], g["sub_" + k]) for k in ("X_sc", "v_s", "a_s", "qdd", "I_s")}
    err["S_s"] = relerr(fields["S_s"][:, dofs], g["sub_S_s"][:, dofs]) if dofs else 0.0
    err["f_tot"] = relerr(fields["ft_s"], g["sub_f_s"] + g["sub_ft_s"])
    return err


# stated bounds (fp32, one substep): kinematic quantities 1e-5 (measured <= 1.4e-6), forces / accelerations 1e-4 (<= 2.9e-5)
BOUNDS = {"X_sc": 1e-5, "S_s": 1e-5, "v_s": 1e-5, "a_s": 1e-5, "I_s": 1e-5, "f_tot": 1e-4, "qdd": 1e-4}
