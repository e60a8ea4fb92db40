"""CPU-only, property-based: the index tables that dsim_build_layout (diffrl_amd/csrc/dsim_layout.hpp) derives from a
kinematic tree -- ancestor chains, subtrees, children, per-body / per-subtree contact lists, ancestor-dof lists, the dof
relation matrix, tree levels, the pre-order `ranges` flag -- against a direct Python computation, for random trees."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from diffrl_amd import dflex as df
from emu_lib import substep_image


@st.composite
def trees(draw):
    L = draw(st.integers(1, 12))
    parents = [-1] + [draw(st.integers(0, i - 1)) for i in range(1, L)]
    kinds = [draw(st.sampled_from(["free", "rev"]))] + [draw(st.sampled_from(["rev", "pri", "ball"])) for _ in range(1, L)]
    shapes = [draw(st.sampled_from(["none", "sphere", "capsule"])) for _ in range(L)]
    return parents, kinds, shapes


def _build(parents, kinds, shapes):
    b = df.sim.ModelBuilder()
    b.add_articulation()
    J = {"free": df.JOINT_FREE, "rev": df.JOINT_REVOLUTE, "pri": df.JOINT_PRISMATIC, "ball": df.JOINT_BALL}
    for i, (p, k, s) in enumerate(zip(parents, kinds, shapes)):
        link = b.add_link(p, df.transform((0.1 * i, 0.0, 0.0), df.quat_identity()), (0.0, 0.0, 1.0), J[k], armature=0.05)
        # every link needs mass for a well-posed model: a tiny sphere when it has no contact shape
        if s == "capsule":
            b.add_shape_capsule(link, radius=0.05, half_width=0.1)
        else:
            b.add_shape_sphere(link, radius=0.05)
    m = b.finalize("cpu")
    m.ground = True
    m.gravity = (0.0, -9.81, 0.0)
    m.collide()
    return m.template()


@settings(max_examples=40, deadline=None)
@given(trees())
def test_tables_match_a_direct_computation(tree):
    parents, kinds, shapes = tree
    t = _build(parents, kinds, shapes)
    L, nd, C = t.n_links, t.n_qd, t.n_contacts
    img, off, dims = substep_image(t, t.joint_q0[None], np.zeros((1, nd), np.float32), np.zeros((1, nd), np.float32), None,
                                   1.0 / 960.0)
    I = img.view(np.int32)

    def csr(start, lst, i):
        a, b = I[off[start] + i], I[off[start] + i + 1]
        return I[off[lst] + a:off[lst] + b].tolist()

    anc = []
    for i in range(L):
        chain, j = [], i
        while j >= 0:
            chain.append(j)
            j = parents[j]
        anc.append(chain[::-1])
    sub = [sorted(j for j in range(L) if i in anc[j]) for i in range(L)]
    qds = list(t.joint_qd_start)
    dofs = [list(range(qds[i], qds[i + 1])) for i in range(L)]
    cbody = list(t.contact_body)
    for i in range(L):
        assert csr("anc_start", "anc_list", i) == anc[i]
        assert csr("sub_start", "sub_list", i) == sub[i]
        assert csr("child_start", "child_list", i) == [j for j in range(L) if parents[j] == i]
        assert csr("cb_start", "cb_list", i) == [k for k in range(C) if cbody[k] == i]
        assert csr("scb_start", "scb_list", i) == [k for k in range(C) if cbody[k] in sub[i]]
        assert csr("adof_start", "adof_list", i) == [d for j in anc[i] for d in dofs[j]]
        li = I[off["linfo"] + 8 * i:off["linfo"] + 8 * i + 8].tolist()
        assert li[0] == parents[i] and li[4] == len(anc[i]) - 1 and li[5] == len(sub[i])
        assert li[7] == len([k for k in range(C) if cbody[k] in sub[i]])
    link_of = [i for i in range(L) for _ in dofs[i]]
    rel = I[off["rel"]:off["rel"] + nd * nd].reshape(nd, nd)
    for a in range(nd):
        for b in range(nd):
            la, lb = link_of[a], link_of[b]
            want = 1 if lb in sub[la] else (2 if la in sub[lb] else 0)
            assert rel[a, b] == want, (a, b)
    preorder = all(sub[i] == list(range(i, i + len(sub[i]))) for i in range(L))
    contiguous = all((lambda ks: ks == list(range(ks[0], ks[0] + len(ks))) if ks else True)(
        [k for k in range(C) if cbody[k] in sub[i]]) for i in range(L))
    assert bool(dims["flags"] & 1) == (preorder and contiguous)
    assert dims["D"] == max(len(c) for c in anc)
    # bounds of the mass-matrix adjoint's two lists per dof (dsim_bwd_mass_bounded) and the eligibility of the fused joint-space
    # adjoint (dsim_bwd_joint_wave): round 4
    assert dims["SDMAX"] == max(sum(len(dofs[j]) for j in sub[i] if j != i) for i in range(L))
    assert dims["ADMAX"] == max(sum(len(dofs[j]) for j in anc[i]) for i in range(L))
    hinge = [kinds[i] in ("rev", "pri") and len(dofs[i]) == 1 for i in range(L)]
    jw = all(hinge[i] or (i == 0 and kinds[0] == "free") for i in range(L))
    assert dims["JW_OK"] == int(jw) and dims["JW_FREE_ROOT"] == int(jw and kinds[0] == "free")


@st.composite
def deep_trees(draw):
    """humanoid-like trees, numbered in pre-order: a trunk chain of 1..4 links below a free root, limbs (chains of 1..6
    hinged links) attached to trunk links -- what the trunk decomposition is for; L >= 11"""
    trunk_len = draw(st.integers(1, 4))
    children = {0: []}
    nodes = [0]
    for k in range(1, trunk_len):
        children[nodes[-1]].append(len(children))
        children[len(children)] = []
        nodes.append(len(children) - 1)
    n_limbs = draw(st.integers(2, 5))
    for _ in range(n_limbs):
        at = draw(st.sampled_from(nodes))
        if len(children[at]) >= 4:
            continue
        prev = at
        for _k in range(draw(st.integers(1, 6))):
            new = len(children)
            children[prev].append(new)
            children[new] = []
            prev = new
    # pre-order relabelling
    order, parent_of = [], {0: -1}
    stack = [0]
    while stack:
        x = stack.pop()
        order.append(x)
        for c in reversed(children[x]):
            parent_of[c] = x
            stack.append(c)
    new_id = {x: i for i, x in enumerate(order)}
    parents = [(-1 if parent_of[x] < 0 else new_id[parent_of[x]]) for x in order]
    L = len(parents)
    while L < 11:   # pad with one more limb off the root so that the decomposition applies (L > 10)
        parents.append(0 if L == len(order) else L - 1)
        L += 1
    shapes = [draw(st.sampled_from(["sphere", "capsule"])) for _ in range(L)]
    return parents, ["free"] + ["rev"] * (L - 1), shapes


@settings(max_examples=40, deadline=None)
@given(deep_trees())
def test_trunk_decomposition_of_deep_trees(tree):
    """DsimDims::NT... (dsim_layout.hpp): the trunk is the ancestor-closed set of links with subtrees above the cap, its
    child / contact / dof records are those of the tree, every other link fits the light caps, light_list lists them."""
    from emu_lib import layout
    parents, kinds, shapes = tree
    t = _build(parents, kinds, shapes)
    L, C = t.n_links, t.n_contacts
    off, d = layout(t)
    img, off2, _ = substep_image(t, t.joint_q0[None], np.zeros((1, t.n_qd), np.float32), np.zeros((1, t.n_qd), np.float32), None,
                                 1.0 / 960.0)
    I = img.view(np.int32)
    sub = [[j for j in range(L) if i in _chain(parents, j)] for i in range(L)]
    cbody = list(t.contact_body)
    scb = [[k for k in range(C) if cbody[k] in sub[i]] for i in range(L)]
    _TREES_SEEN.append(1)
    if d["NT"] == 0:
        return   # the builder found no admissible cap (too many trunk links / children / contacts): flat sums are used
    _TRUNK_SEEN.append(1)
    nt, cap = d["NT"], d["LCAP"]
    trunk = d["trunk"][:nt]
    assert trunk == [i for i in range(L) if len(sub[i]) > cap] and trunk[0] == 0 and 1 <= nt <= 6
    assert d["NLT"] == L - nt
    light = [i for i in range(L) if i not in trunk]
    assert I[off2["light_list"]:off2["light_list"] + len(light)].tolist() == light
    assert max(len(sub[i]) for i in light) <= cap <= 8
    assert d["CCAP"] == max([len(scb[i]) for i in light] + [0]) <= 8
    qds = list(t.joint_qd_start)
    for u, i in enumerate(trunk):
        assert parents[i] == (-1 if d["tr_par"][u] < 0 else trunk[d["tr_par"][u]])
        ch = [j for j in range(L) if parents[j] == i]
        assert d["tr_nch"][u] == len(ch) <= 4 and d["tr_ch"][4 * u:4 * u + len(ch)] == ch
        own = [k for k in range(C) if cbody[k] == i]
        assert d["tr_ncb"][u] == len(own) and (not own or d["tr_cb0"][u] == own[0])
        assert (d["tr_d0"][u], d["tr_nd"][u]) == (qds[i], qds[i + 1] - qds[i])
    # trunk + light subtrees tile the tree: a trunk link's subtree = itself + the subtrees of its children
    for i in trunk:
        kids = [j for j in range(L) if parents[j] == i]
        assert sorted([i] + [x for j in kids for x in sub[j]]) == sub[i]


_TRUNK_SEEN = []
_TREES_SEEN = []


def test_trunk_decomposition_was_exercised():
    if not _TREES_SEEN:   # (pytest-xdist may hand the property test to another worker process)
        pytest.skip("the deep-tree property test did not run in this process")
    assert len(_TRUNK_SEEN) >= 5, "the random deep trees should admit a trunk decomposition most of the time"


def _chain(parents, j):
    out = []
    while j >= 0:
        out.append(j)
        j = parents[j]
    return out


def test_muscle_chunk_tables_cover_the_body_rows():
    """mc_row / mc_cnt / mb_start (dsim_layout.hpp): the chunks of a body tile its muscle wrench rows, at most DSIM_MUSCLE_CHUNK = 16
    rows each, in the padded row numbering of `mus`; seg_rec: the packed per-segment records against the tables they are built from,
    recomputed here from the template (they are not in the image: no kernel reads them)"""
    from emu_lib import layout
    from oracle_lib import template_from_golden
    t = template_from_golden("snu")
    off, d = layout(t)
    img, off2, _ = substep_image(t, t.joint_q0[None], np.zeros((1, t.n_qd), np.float32), np.zeros((1, t.n_qd), np.float32),
                                 np.zeros((1, t.n_muscles), np.float32), 1.0 / 2880.0)
    I = img.view(np.int32)
    L, K, NS = t.n_links, d["MK"], d["NS"]
    assert K > 0
    for gone in ("seg_wp", "seg_m", "ml_start", "ml_list", "seg_slot", "mlinks"):
        assert gone not in off2
    # active segments (two consecutive waypoints of a muscle on different links), in muscle order; per link the list of its segment ends
    links, ms = np.asarray(t.muscle_links), np.asarray(t.muscle_start)
    wp = np.array([w for m in range(t.n_muscles) for w in range(ms[m], ms[m + 1] - 1) if links[w] != links[w + 1]])
    sm = np.array([m for m in range(t.n_muscles) for w in range(ms[m], ms[m + 1] - 1) if links[w] != links[w + 1]])
    assert len(wp) == NS
    np.testing.assert_array_equal(I[off2["ms_start"]:off2["ms_start"] + t.n_muscles + 1],
                                  np.searchsorted(sm, np.arange(t.n_muscles + 1)))
    ends = [[] for _ in range(L)]   # link -> [2 * segment + side], by segment
    for s_, w in enumerate(wp):
        ends[links[w]].append(2 * s_)
        ends[links[w + 1]].append(2 * s_ + 1)
    mb = I[off2["mb_start"]:off2["mb_start"] + L + 1].tolist()
    row, cnt = I[off2["mc_row"]:off2["mc_row"] + K].tolist(), I[off2["mc_cnt"]:off2["mc_cnt"] + K].tolist()
    assert mb[0] == 0 and mb[L] == K
    # chunk e owns the 16 rows from row 17 e on (DSIM_MUSCLE_STRIDE: 102 words, so that the (chunk, component) lanes of a wavefront
    # read different banks); the first cnt[e] of them are the rows of the body's segment ends, in the order of the body's list, the
    # rest are never written (zeros: a chunk sum is 16 unconditional additions)
    assert row == [17 * e for e in range(K)]
    slot = np.full(2 * NS, -1)
    for i in range(L):
        rows = [r for e in range(mb[i], mb[i + 1]) for r in range(row[e], row[e] + cnt[e])]
        assert len(rows) == len(ends[i]), "body %d" % i
        slot[ends[i]] = rows
        assert all(0 < cnt[e] <= 16 for e in range(mb[i], mb[i + 1]))
        assert all(cnt[e] == 16 for e in range(mb[i], mb[i + 1] - 1)), "only a body's last chunk is filled up"
    assert (slot >= 0).all() and len(set(slot.tolist())) == 2 * NS
    assert off2["mpart"] - off2["mus"] >= 6 * 17 * K + NS, "rows of all chunks + the activation cotangents behind them"
    assert 6 * K <= 192, "one pass of the (chunk, component) items over the three wavefronts behind the first one"
    rec = I[off2["seg_rec"]:off2["seg_rec"] + 8 * NS].reshape(NS, 8)
    assert off2["seg_rec"] % 4 == 0
    np.testing.assert_array_equal(rec[:, 0], 7 * links[wp])
    np.testing.assert_array_equal(rec[:, 1], 7 * links[wp + 1])
    np.testing.assert_array_equal(rec[:, 2], 3 * wp)
    np.testing.assert_array_equal(rec[:, 3], sm)
    np.testing.assert_array_equal(rec[:, 4], 6 * slot[0::2])
    np.testing.assert_array_equal(rec[:, 5], 6 * slot[1::2])


def test_every_array_of_the_shipped_models_is_within_the_lds_offset_field():
    """An LDS instruction addresses base register + 16-bit byte offset.  The specialised kernels know every array's offset at compile
    time; an array that starts beyond 64 KiB needs its address in a register of its own (round 5: SNUHumanoid's aH / gua / agx did --
    the adjoint kernel sat at 256 VGPRs with two spilled once the image grew by another 700 words; without 1,836 words of tables that
    no kernel read everything is below the limit again and the launch is 2 % shorter)"""
    from diffrl_amd import specialise
    for tag, t in specialise.shipped_templates():
        off, dims = specialise.layout(t)
        last = max(v for k, v in off.items() if k not in ("total_words", "fwd_words", "save_words", "const_words") and v >= 0)
        assert 4 * last < 65536, (tag, last)
        assert 4 * off["total_words"] <= 65536 + 4 * 384, (tag, off["total_words"])   # (the spare tail behind the last array may cross it)
