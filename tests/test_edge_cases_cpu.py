"""CPU-only edge cases: model validation in the C-ABI library (runs before any HIP call, so it works without a
GPU), layout limits, ragged / degenerate articulations through the oracle and the lane-serial phase harness."""
import ctypes as C

import numpy as np
import pytest

from diffrl_amd import capi
from diffrl_amd import dflex as df
from emu_lib import emu_backward, emu_forward, layout
from oracle_lib import golden, oracle_backward, project_tangent, relerr, step_grad_tolerance, template_from_golden


def _create(t):
    lib = capi.lib()
    desc, keep = capi.make_desc(t)
    h = C.c_void_p()
    rc = lib.dsim_model_create(C.byref(desc), C.byref(h))
    return rc, lib.dsim_last_error().decode(), h


def test_rejects_non_block_diagonal_inertia():
    t = template_from_golden("ant")
    t.body_I_m = t.body_I_m.copy()
    t.body_I_m[2, 0, 4] = 0.1
    rc, msg, _ = _create(t)
    assert rc == -1 and "block diagonal" in msg


def test_rejects_child_before_parent_and_bad_counts():
    t = template_from_golden("cartpole")
    t.joint_parent = t.joint_parent.copy()
    t.joint_parent[1] = 2
    rc, msg, _ = _create(t)
    assert rc == -1 and "parent" in msg
    t = template_from_golden("cartpole")
    t.joint_type = t.joint_type.copy()
    t.joint_type[1] = 4  # claims a free joint but has 1 coordinate
    rc, msg, _ = _create(t)
    assert rc == -1 and "type" in msg


def test_rejects_rotated_com_frame_and_out_of_range_contact():
    t = template_from_golden("ant")
    t.joint_X_cm = t.joint_X_cm.copy()
    t.joint_X_cm[1, 3:7] = (0.0, 0.7071068, 0.0, 0.7071068)
    rc, msg, _ = _create(t)
    assert rc == -1 and "joint_X_cm" in msg
    t = template_from_golden("ant")
    t.contact_body = t.contact_body.copy()
    t.contact_body[0] = 99
    t.validate = lambda: None
    rc, msg, _ = _create(t)
    assert rc == -1 and "contact_body" in msg


def test_null_arguments():
    lib = capi.lib()
    assert lib.dsim_model_create(None, None) == -1
    assert lib.dsim_model_destroy(None) == 0
    assert lib.dsim_ckpt_floats(None, 4) == 0


def _chain(n_links, with_shapes=True, floating=False):
    """user-built articulation through the ModelBuilder API: a revolute chain hanging under gravity"""
    b = df.sim.ModelBuilder()
    b.add_articulation()
    parent = -1
    for i in range(n_links):
        kind = df.JOINT_FREE if (floating and i == 0) else df.JOINT_REVOLUTE
        link = b.add_link(parent, df.transform((0.0 if i == 0 else 0.4, 0.0, 0.0), df.quat_identity()), (0.0, 0.0, 1.0),
                          kind, stiffness=1.0, damping=0.2, limit_lower=-1.0, limit_upper=1.0, armature=0.02)
        if with_shapes or i == n_links - 1:
            b.add_shape_capsule(link, pos=(0.2, 0.0, 0.0), radius=0.05, half_width=0.2, ke=1e4, kd=1e3, kf=1e3, mu=0.5)
        parent = link
    if floating:
        b.joint_q[0:3] = [0.0, 0.3, 0.0]
    m = b.finalize("cpu")
    m.ground = True
    m.gravity = (0.0, -9.81, 0.0)
    m.collide()
    return m.template()


@pytest.mark.parametrize("n_links,shapes,floating", [(1, True, False), (5, False, False), (4, True, True)])
def test_user_built_chain_emulated_kernels_vs_oracle(n_links, shapes, floating):
    """single link, mass-less intermediate links (ragged mass distribution), floating chain in contact"""
    t = _chain(n_links, shapes, floating)
    off, dims = layout(t)
    assert dims["L"] == n_links
    rng = np.random.default_rng(n_links)
    n = 3
    q = np.tile(t.joint_q0, (n, 1)) + rng.normal(0, 0.2, (n, t.n_q)).astype(np.float32)
    if floating:
        q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    qd = rng.normal(0, 0.5, (n, t.n_qd)).astype(np.float32)
    act = rng.normal(0, 1.0, (n, t.n_qd)).astype(np.float32)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    for S, mm in [(1, 1), (5, 2), (3, 7)]:     # mm_freq larger than / not dividing the substep count
        dt = S / 960.0                          # substep length of the shipped environments (1/60 / 16)
        o = oracle_backward(t, q, qd, act, None, dt, S, mm, gq, gqd)
        qo, qdo, ck = emu_forward(t, q, qd, act, None, dt, S, mm, want_ckpt=True)
        r = emu_backward(t, ck, act, None, dt, S, mm, gq, gqd)
        # mass-less intermediate links make H nearly singular (only the armature regularises it): the reference's
        # Cholesky + substitution and the explicit inverse used here are both at the fp32 noise floor of that system
        tol = 2e-5 if shapes else 5e-4
        assert relerr(qo, o["q_out"]) < tol and relerr(qdo, o["qd_out"]) < 10 * tol
        gtol = 1e-3 if shapes else 2e-2
        assert relerr(project_tangent(t, q, r["gq"]), project_tangent(t, q, o["gq"])) < gtol
        assert relerr(r["gqd"], o["gqd"]) < gtol and relerr(r["gact"], o["gact"]) < gtol


def test_degenerate_inputs_stay_finite():
    """zero velocities / zero actions / resting contact with exactly zero tangential velocity: the reference's
    zero-gradient rules for normalize / length at 0 (vec3.h:204-222) keep everything finite"""
    t = template_from_golden("ant")
    g = golden("ant_step")
    q = g["q_in"][8:10].copy()
    qd = np.zeros((2, t.n_qd), np.float32)
    act = np.zeros((2, t.n_qd), np.float32)
    S, mm, dt = 16, 16, 1 / 60
    qo, qdo, ck = emu_forward(t, q, qd, act, None, dt, S, mm, want_ckpt=True)
    r = emu_backward(t, ck, act, None, dt, S, mm, np.ones_like(q), np.ones_like(qd))
    o = oracle_backward(t, q, qd, act, None, dt, S, mm, np.ones_like(q), np.ones_like(qd))
    for a in (qo, qdo, r["gq"], r["gqd"], r["gact"]):
        assert np.isfinite(a).all()
    assert relerr(qo, o["q_out"]) < 2e-5
    assert relerr(r["gact"], o["gact"]) < 1e-3


def _random_tree(seed, floating, muscles=0):
    """random articulation: branching tree numbered breadth-first (NOT pre-order, so subtrees / contact sets are not
    contiguous ranges and the CSR-list code paths run), mixed revolute / prismatic / ball joints, rotated joint frames,
    spheres / capsules / boxes with ground contacts"""
    rng = np.random.default_rng(seed)
    L = int(rng.integers(6, 10))
    b = df.sim.ModelBuilder()
    b.add_articulation()
    parents = [-1] + [int(rng.integers(0, max(1, (i + 1) // 2))) for i in range(1, L)]   # breadth-first-ish: small parent ids
    for i in range(L):
        if i == 0:
            kind = df.JOINT_FREE if floating else df.JOINT_REVOLUTE
        else:
            kind = [df.JOINT_REVOLUTE, df.JOINT_PRISMATIC, df.JOINT_BALL, df.JOINT_REVOLUTE][int(rng.integers(0, 4))]
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        rot = rng.normal(size=4)
        rot /= np.linalg.norm(rot)
        pos = (0.0, 0.0, 0.0) if i == 0 else tuple(rng.uniform(-0.3, 0.3, 3))
        link = b.add_link(parents[i], df.transform(pos, tuple(rot)), tuple(axis), kind, stiffness=float(rng.uniform(0, 2)),
                          damping=float(rng.uniform(0.05, 0.5)), limit_lower=-0.8, limit_upper=0.8, armature=0.02)
        shape = int(rng.integers(0, 3))
        kw = dict(ke=1e4, kd=1e3, kf=1e3, mu=float(rng.uniform(0.3, 1.0)))
        if shape == 0:
            b.add_shape_sphere(link, pos=tuple(rng.uniform(-0.1, 0.1, 3)), radius=0.08, **kw)
        elif shape == 1:
            b.add_shape_capsule(link, pos=(0.1, 0.0, 0.0), radius=0.05, half_width=0.12, **kw)
        else:
            b.add_shape_box(link, pos=(0.0, 0.05, 0.0), hx=0.08, hy=0.05, hz=0.06, **kw)
    for _ in range(muscles):
        # way-points on random links, consecutive ones sometimes on the SAME link (such segments exert no force)
        k = int(rng.integers(3, 6))
        links = [int(x) for x in rng.integers(0, L, k)]
        b.add_muscle(links, [tuple(rng.uniform(-0.1, 0.1, 3)) for _ in range(k)], 1.0, 0.1, 0.1, 0.2, 0.0)
    if floating:
        b.joint_q[0:3] = [0.0, 0.25, 0.0]
    m = b.finalize("cpu")
    m.ground = True
    m.gravity = (0.0, -9.81, 0.0)
    m.collide()
    return m.template(), parents


def _tree_states(t, rng, n):
    q = np.tile(t.joint_q0, (n, 1)) + rng.normal(0, 0.15, (n, t.n_q)).astype(np.float32)
    for i in range(t.n_links):
        ty, cs = int(t.joint_type[i]), int(t.joint_q_start[i])
        sl = slice(cs + 3, cs + 7) if ty == df.JOINT_FREE else (slice(cs, cs + 4) if ty == df.JOINT_BALL else None)
        if sl is not None:
            q[:, sl] /= np.linalg.norm(q[:, sl], axis=1, keepdims=True)
    qd = rng.normal(0, 0.5, (n, t.n_qd)).astype(np.float32)
    act = rng.normal(0, 1.0, (n, t.n_qd)).astype(np.float32)
    return q.astype(np.float32), qd, act


@pytest.mark.parametrize("seed,floating", [(0, True), (1, False), (2, True), (3, True), (4, False)])
def test_random_trees_emulated_kernels_vs_oracle(seed, floating):
    t, parents = _random_tree(seed, floating)
    off, dims = layout(t)
    if seed in (0, 2):
        assert dims["flags"] & 1 == 0, "these trees are meant to exercise the non-contiguous (CSR list) code paths"
    rng = np.random.default_rng(100 + seed)
    q, qd, act = _tree_states(t, rng, 3)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    for S, mm in [(4, 2), (3, 3)]:
        dt = S / 960.0
        o = oracle_backward(t, q, qd, act, None, dt, S, mm, gq, gqd)
        qo, qdo, ck = emu_forward(t, q, qd, act, None, dt, S, mm, want_ckpt=True)
        r = emu_backward(t, ck, act, None, dt, S, mm, gq, gqd)
        assert relerr(qo, o["q_out"]) < 5e-5 and relerr(qdo, o["qd_out"]) < 5e-4
        err = dict(gq=relerr(project_tangent(t, q, r["gq"]), project_tangent(t, q, o["gq"])), gqd=relerr(r["gqd"], o["gqd"]),
                   gact=relerr(r["gact"], o["gact"]))
        tol = step_grad_tolerance(t, q, qd, act, None, dt, S, mm, gq, gqd, err, ref=o)   # 1e-3, or probed
        assert all(err[k] < tol[k] for k in err), (err, tol)


@pytest.mark.parametrize("seed,floating", [(5, True), (6, False)])
def test_random_trees_with_muscles(seed, floating):
    """line-of-action muscles (eval_muscles, sim.py:1209-1265) routed over random links of a breadth-first-numbered tree"""
    t, parents = _random_tree(seed, floating, muscles=4)
    assert t.n_muscles == 4
    rng = np.random.default_rng(200 + seed)
    q, qd, act = _tree_states(t, rng, 3)
    mact = rng.uniform(0.0, 30.0, (3, t.n_muscles)).astype(np.float32)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    S, mm = 4, 2
    dt = S / 960.0
    o = oracle_backward(t, q, qd, act, mact, dt, S, mm, gq, gqd)
    qo, qdo, ck = emu_forward(t, q, qd, act, mact, dt, S, mm, want_ckpt=True)
    r = emu_backward(t, ck, act, mact, dt, S, mm, gq, gqd)
    assert relerr(qo, o["q_out"]) < 5e-5 and relerr(qdo, o["qd_out"]) < 5e-4
    err = dict(gq=relerr(project_tangent(t, q, r["gq"]), project_tangent(t, q, o["gq"])), gqd=relerr(r["gqd"], o["gqd"]),
               gact=relerr(r["gact"], o["gact"]), gmact=relerr(r["gmact"], o["gmact"]))
    tol = step_grad_tolerance(t, q, qd, act, mact, dt, S, mm, gq, gqd, err, ref=o)   # 1e-3, or probed
    assert np.abs(o["gmact"]).max() > 0 and all(err[k] < tol[k] for k in err), (err, tol)


def test_random_tree_with_many_muscles_fills_several_chunks_per_body():
    """40 muscles over 6-9 links: bodies with more than DSIM_MUSCLE_CHUNK = 16 segment ends, i.e. several chunks of muscle rows per
    body, the last one of each filled up with rows nothing writes (dsim_layout.hpp: seg_slot / mc_row) -- the run-time-layout kernels
    sum the rows by their counts, so a wrong row number or a fill row that is not zero shows up against the scalar oracle"""
    t, parents = _random_tree(7, True, muscles=40)
    off, dims = layout(t)
    from emu_lib import substep_image
    img, off2, _ = substep_image(t, t.joint_q0[None], np.zeros((1, t.n_qd), np.float32), np.zeros((1, t.n_qd), np.float32),
                                 np.zeros((1, t.n_muscles), np.float32), 1.0 / 960.0)
    I = img.view(np.int32)
    mb = I[off2["mb_start"]:off2["mb_start"] + t.n_links + 1]
    cnt = I[off2["mc_cnt"]:off2["mc_cnt"] + dims["MK"]]
    assert np.diff(mb).max() >= 2 and cnt.min() < 16 and cnt.max() == 16, "several chunks on one body, and chunks that are filled up"
    assert cnt.sum() == 2 * dims["NS"]
    rng = np.random.default_rng(207)
    q, qd, act = _tree_states(t, rng, 2)
    mact = rng.uniform(0.0, 10.0, (2, t.n_muscles)).astype(np.float32)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    S, mm = 2, 2
    dt = S / 960.0
    o = oracle_backward(t, q, qd, act, mact, dt, S, mm, gq, gqd)
    qo, qdo, ck = emu_forward(t, q, qd, act, mact, dt, S, mm, want_ckpt=True)
    r = emu_backward(t, ck, act, mact, dt, S, mm, gq, gqd)
    assert relerr(qo, o["q_out"]) < 5e-5 and relerr(qdo, o["qd_out"]) < 5e-4
    err = dict(gq=relerr(project_tangent(t, q, r["gq"]), project_tangent(t, q, o["gq"])), gqd=relerr(r["gqd"], o["gqd"]),
               gact=relerr(r["gact"], o["gact"]), gmact=relerr(r["gmact"], o["gmact"]))
    tol = step_grad_tolerance(t, q, qd, act, mact, dt, S, mm, gq, gqd, err, ref=o)   # 1e-3, or probed
    assert np.abs(o["gmact"]).max() > 0 and all(err[k] < tol[k] for k in err), (err, tol)


def test_half_angle_polynomials_at_and_beyond_their_range():
    """dsim_math.hpp::half_angle_sincos replaces sinf / cosf of a joint half-angle by degree-11 / 12 polynomials for
    |angle / 2| <= pi / 2 and falls back to the library routines beyond.  Isolated here: a revolute chain whose joint angles sit
    at the ends of the hot range, just inside and just outside the switch (|q| = pi -/+ 1e-3), at +-pi exactly and far outside
    (several turns), first-substep link poses X_sc and the end state against the scalar oracle, which calls sinf / cosf as the
    reference does (quat.h:44-52).  Bound: 1e-6 relative on the poses -- below the 1e-5 budget of a substep by a decade."""
    from ckpt_fields import first_substep
    from oracle_lib import oracle_forward
    t = _chain(3, with_shapes=True, floating=False)
    angles = np.array([0.0, 1.0, -1.0, np.pi - 1e-3, -(np.pi - 1e-3), np.pi, -np.pi, np.pi + 1e-3, -(np.pi + 1e-3), 2.5 * np.pi,
                       -7.3, 3.0, -3.1], np.float32)
    n = len(angles)
    q = np.zeros((n, t.n_q), np.float32)
    q[:, 0] = angles
    q[:, 1] = angles[::-1]
    q[:, 2] = 0.5 * angles
    qd = np.zeros((n, t.n_qd), np.float32)
    act = np.zeros((n, t.n_qd), np.float32)
    dt, S, mm = 1.0 / 960.0, 1, 1
    qo, qdo, ck = emu_forward(t, q, qd, act, None, dt, S, mm, want_ckpt=True)
    o_q, o_qd, dbg = oracle_forward(t, q, qd, act, None, dt, S, mm, debug=True)
    X = first_substep(t, ck)["X_sc"]
    assert relerr(X, dbg["X_sc"]) < 1e-6
    assert relerr(qo, o_q) < 1e-6


def _caterpillar():
    """A pre-order tree of 17 links whose ROW-TREE step list (dsim_layout.hpp: DsimDims::rt_kind) mixes all three kinds and puts a
    FAR step in front of row-shift steps: a spine 0-2-4-...-14 (each spine link the child of the one two below it), a leaf on every
    spine link (1, 3, ..., 13: distance 1), and two leaves on the last one -- 15 (distance 1) and 16 (distance 2 ACROSS the 16-lane
    row boundary: no row shift reaches from lane 16 to lane 14, so the layout builder emits a FAR step, by v_readlane, at the
    deepest level, followed by the wave / row shifts of the shallower levels).  Floating base, hinges in rotated frames, capsules
    with ground contacts.  (The shipped models have FAR steps only at the END of their lists; the s_nop hazard guard in front of
    the inline-asm DPP steps must also cover a DPP step that follows compiler-generated v_readlane + v_fma code.)"""
    rng = np.random.default_rng(11)
    b = df.sim.ModelBuilder()
    b.add_articulation()
    parents = [-1]
    for i in range(1, 17):
        if i in (15, 16):
            parents.append(14)
        elif i % 2 == 0:
            parents.append(i - 2)   # spine
        else:
            parents.append(i - 1)   # leaf of the spine link below it
    for i in range(17):
        kind = df.JOINT_FREE if i == 0 else df.JOINT_REVOLUTE
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        rot = rng.normal(size=4)
        rot /= np.linalg.norm(rot)
        pos = (0.0, 0.0, 0.0) if i == 0 else ((0.22, 0.0, 0.0) if i % 2 == 0 else tuple(rng.uniform(-0.15, 0.15, 3)))
        link = b.add_link(parents[i], df.transform(pos, tuple(rot)), tuple(axis), kind, stiffness=float(rng.uniform(0, 2)),
                          damping=float(rng.uniform(0.05, 0.5)), limit_lower=-0.8, limit_upper=0.8, armature=0.02)
        b.add_shape_capsule(link, pos=(0.08, 0.0, 0.0), radius=0.04, half_width=0.08, ke=1e4, kd=1e3, kf=1e3, mu=float(rng.uniform(0.3, 1.0)))
    b.joint_q[0:3] = [0.0, 0.2, 0.0]
    m = b.finalize("cpu")
    m.ground = True
    m.gravity = (0.0, -9.81, 0.0)
    m.collide()
    return m.template(), parents


def test_caterpillar_row_tree_has_a_far_step_in_front_of_shift_steps():
    t, parents = _caterpillar()
    off, d = layout(t)
    n = d["RT_N"]
    kinds = d["rt_kind"][:n]
    assert d["flags"] & 1 and d["L"] == 17 and n > 0
    FAR, ROW, WAVE1 = 2, 0, 1
    assert FAR in kinds and ROW in kinds and WAVE1 in kinds
    far_at = [k for k in range(n) if kinds[k] == FAR]
    assert any(k + 1 < n and kinds[k + 1] in (ROW, WAVE1) for k in far_at), kinds   # the case the hazard guard must cover
    assert any(d["rt_d"][k] == 16 and d["rt_lvl"][k] == 14 for k in far_at)           # child lane 16 -> parent lane 14


def test_caterpillar_emulated_kernels_vs_oracle():
    """the generic phase code on the caterpillar (forward + adjoint) against the scalar oracle"""
    t, parents = _caterpillar()
    rng = np.random.default_rng(21)
    q, qd, act = _tree_states(t, rng, 1)
    gq, gqd = rng.normal(0, 1, q.shape).astype(np.float32), rng.normal(0, 1, qd.shape).astype(np.float32)
    S, mm = 2, 2
    dt = S / 960.0
    o = oracle_backward(t, q, qd, act, None, dt, S, mm, gq, gqd)
    qo, qdo, ck = emu_forward(t, q, qd, act, None, dt, S, mm, want_ckpt=True)
    r = emu_backward(t, ck, act, None, dt, S, mm, gq, gqd)
    assert relerr(qo, o["q_out"]) < 5e-5 and relerr(qdo, o["qd_out"]) < 5e-4
    err = dict(gq=relerr(project_tangent(t, q, r["gq"]), project_tangent(t, q, o["gq"])), gqd=relerr(r["gqd"], o["gqd"]),
               gact=relerr(r["gact"], o["gact"]))
    tol = step_grad_tolerance(t, q, qd, act, None, dt, S, mm, gq, gqd, err, ref=o)
    assert all(err[k] < tol[k] for k in err), (err, tol)
