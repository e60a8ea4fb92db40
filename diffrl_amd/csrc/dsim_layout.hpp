// dsim_layout.hpp -- host-side analysis of one articulation template and the LDS image layout.
//
// `dsim_model_create` (include/dsim.h) turns the reference-style flat model arrays
// (dflex/dflex/model.py:1646-1879) into ONE packed constant block that every workgroup copies into
// LDS at kernel start, followed by the per-environment work arrays.  All offsets are in 32-bit
// words from the start of the workgroup's dynamic LDS segment and are passed to the kernels by
// value (they live in SGPRs).
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dsim.h"

#define DSIM_PMASK_N 10
#define DSIM_TRUNK_MAX 6
#define DSIM_TRUNK_CH 4
#define DSIM_LIGHT_CAP 8    // register budget of the light sums: LCAP, CCAP <= this
#define DSIM_MUSCLE_STRIDE 17  // rows from one chunk to the next in `mus` (see seg_slot below)
#define DSIM_MUSCLE_CHUNK 16   // rows per chunk of the per-body muscle-row gather (SNUHumanoid: 396 rows in 30 chunks -- 180 (chunk, component) items,
                               // one pass over the 192 lanes that sum them beside the first wavefront's link-level work)
#define DSIM_TAIL_PAD 384
#define DSIM_RT_MAX 16      // steps of a row-tree sum (DsimDims::RT_N)
struct DsimDims {
    int L, nq, nd, C, M, W, NS, D;  // links, coords, dofs, contacts, muscles, waypoints, active muscle segments, tree levels
    int flags;                      // DSIM_F_*
    int tmask;                      // bit t set: some joint has type t
    int pmask[DSIM_PMASK_N];        // bit t set: some link has a joint of type t at position p of its ancestor chain (root = 0)
    // Trunk decomposition of a deep tree (NT == 0: none; dsim_core.hpp: dsim_trunk_sum).  Links whose subtree is larger
    // than LCAP form the "trunk" (an ancestor-closed set around the root); every other ("light") link has a subtree of
    // at most LCAP links / CCAP contacts, summed flat in one bounded pass; a trunk link then adds its own row, its own
    // contacts and the finished sums of its children, deepest trunk link first.
    int NT, NLT, LCAP, CCAP;                       // trunk links, light links, caps of the light sums
    int trunk[DSIM_TRUNK_MAX];                     // ascending (pre-order: parents first)
    int tr_par[DSIM_TRUNK_MAX];                    // position of the parent in trunk[], -1 for the root
    int tr_nch[DSIM_TRUNK_MAX];                    // number of children
    int tr_ch[DSIM_TRUNK_MAX * DSIM_TRUNK_CH];     // their link indices
    int tr_cb0[DSIM_TRUNK_MAX], tr_ncb[DSIM_TRUNK_MAX];  // the link's own contacts [cb0, cb0 + ncb)
    int tr_d0[DSIM_TRUNK_MAX], tr_nd[DSIM_TRUNK_MAX];    // its own dofs [d0, d0 + nd)
    int MK;                                        // chunks of the per-body muscle-row gather (DsimOff::mc_row)
    int pident;                                    // bit p set: every joint at chain position p has an identity X_pj ROTATION
    // Row-tree sums (RT_N == 0: none; dsim_core.hpp: dsim_rowtree_sum).  A tree of at most 32 links sits on the first lanes of
    // the wavefront (link i on lane i), so a subtree sum over per-link values held in REGISTERS is a handful of lane shifts:
    // bottom-up by levels, one step per distinct distance d = child - parent among the links of a level -- every parent adds
    // the (finished) total of its child d lanes above, weighted 1 / 0.  Step s, deepest level first, is of kind rt_kind[s]:
    //   DSIM_RT_ROW    DPP row_shl:d, d = rt_d[s], children of level rt_lvl[s] (every such edge stays inside a 16-lane row)
    //   DSIM_RT_WAVE1  DPP wave_shl:1 (d = 1; crosses row boundaries), children of level rt_lvl[s]
    //   DSIM_RT_FAR    ONE edge that no row shift reaches: child lane rt_d[s] -> parent lane rt_lvl[s], by v_readlane
    // CBMAX: most contacts on one body (the per-body contact gather that feeds those sums).
    int RT_N, CBMAX;
    int rt_lvl[DSIM_RT_MAX], rt_d[DSIM_RT_MAX], rt_kind[DSIM_RT_MAX];
    // Link <-> dof lane shifts of a row tree (DSH_OK): every link but the root has exactly one dof and its index is the link's
    // index + DSH (pre-order trees whose only multi-dof joint is the root: Ant 5, the planar models 0, cartpole -1); the root has
    // ND_ROOT dofs 0 .. ND_ROOT - 1 (0: a fixed root).  A per-link value then reaches its dof's lane by ONE row shift (the root's
    // by v_readlane from lane 0) instead of a ds_bpermute round trip, and back.
    int DSH_OK, DSH, ND_ROOT;
    // bounds of the mass-matrix adjoint's two lists per dof (dsim_core.hpp: dsim_bwd_mass): most dofs in the STRICT subtree of a
    // link, longest list of dofs of the ancestors-or-self of a link
    int SDMAX, ADMAX;
    // fused joint-space adjoint (dsim_core.hpp: dsim_bwd_joint_wave): every link but the root is a hinge / prismatic joint, the root
    // is one too or a free joint (JW_FREE_ROOT)
    int JW_OK, JW_FREE_ROOT;
};
#define DSIM_TM(t) (1 << (t))
#define DSIM_RT_ROW 0
#define DSIM_RT_WAVE1 1
#define DSIM_RT_FAR 2
#define DSIM_F_RANGES 1  // subtree(i) == links [i, i+n_i) and its contacts == one contiguous contact range (pre-order numbering)

struct DsimOff {
    // ---- constant block: ints
    int jtype, parent, qstart, qdstart, dof_link;
    int linfo;                  // [L][8] packed per-link record: parent, type, q start, qd start, level, subtree size, first subtree contact, subtree contact count
    int anc_start, anc_list;    // ancestors-or-self of link i, root first
    int adof_start, adof_list;  // dofs of all ancestors-or-self of link i
    int sub_start, sub_list;    // subtree of link i (self first, then descendants ascending)
    int child_start, child_list;
    int light_list;             // links outside the trunk, ascending (trunk decomposition, DsimDims::NT > 0)
    int cb_start, cb_list;      // contacts of body i
    int scb_start, scb_list;    // contacts of all bodies in subtree(i) (ascending contact index)
    int rel;                    // [nd*nd] 0 unrelated, 1: link(b) in subtree(link(a)), 2: link(a) strictly below link(b)
    int cbody;
    int ms_start;               // muscle -> [first, last) active segment (a segment is active if its two waypoints lie on different links)
    // The muscle wrench rows of a body (one per segment end on it, kept together) are gathered in two steps: chunks of at most
    // DSIM_MUSCLE_CHUNK rows are summed by one lane each (one LDS round trip), then the chunk sums of a body -- a body with 86 rows was
    // 11 dependent round trips of one lane.  mc_row / mc_cnt: first row and row count of chunk e; mb_start: body -> its chunks.
    int mc_row, mc_cnt, mb_start;
    // packed per-segment record [NS][8] (16-byte aligned): 7 * link 0, 7 * link 1 (X_sc offsets), 3 * first waypoint (mpoints offset),
    // muscle index, 6 * row of link 0, 6 * row of link 1 (offsets into `mus`), 0, 0 -- ONE LDS round trip (two 16-byte reads).  The
    // tables it is built from (segment -> waypoint / muscle, waypoint -> link, link -> segment ends, segment end -> row) are not
    // in the image any more: no kernel reads them, and they were 7 KB of SNUHumanoid's LDS image and of every launch's constant load.
    int seg_rec;
    // ---- constant block: floats
    int xpj, com, axis, ic6, mass, tke, tkd, lke, lkd, target, lower, upper, arm;
    int cpoint, cdist, cmat, grav, mpoints;
    int const_words;
    // ---- forward work arrays (floats)
    int q, qd, act, mact, ua, obs, xsc, S, v, a, i10, f, ftot, cw, tau, qdd, ic10, F, hinv, prow, pcol, mus;
    int cwb;                    // [L][6] contact wrenches summed per body (row-tree models: the side block of the kinematics phase)
    int mpart;                  // [MK][6] chunk sums of the muscle-row gather
    int epf;                    // episode flags (fused env surface): [0] invalid state seen, [1] episode finished
    int qil;                    // one spare word of qdd's 16-byte padding INSIDE the saved block (-1: none): 1 / |r + dr h| of the
                                // free root's quaternion update, left there by the integrator for integrate^T (dsim_core.hpp: DsimSavedIl)
    int save_words;             // length of the saved block that starts at q (see dsim_build_layout)
    int fwd_words;
    // ---- adjoint work arrays (floats)
    int aq, aqd, aqn, aqdn, aact, amact, aqdd, atau, aS, af, acx, av, aa, aatot, avtot, avj,
        aw, awp, azs, ai10m, aic10, aH, gua, agx;
    int total_words;
};

// checkpoint geometry (floats per environment): one saved block per substep + one H^-1 per mass-matrix group +
// the tail [q, qd of the step's end state (before any episode reset), episode flags]
inline int dsim_hinv_words(int nd) { return (nd * nd + 3) & ~3; }
inline int dsim_tail_words(int nq, int nd) { return (nq + nd + 1 + 3) & ~3; }
inline long long dsim_ckpt_tail_offset(int save_words, int nd, int substeps, int mm_freq) {
    const int groups = (substeps + mm_freq - 1) / mm_freq;
    return (long long)substeps * save_words + (long long)groups * dsim_hinv_words(nd);
}
inline long long dsim_ckpt_words(int save_words, int nq, int nd, int substeps, int mm_freq) {
    return dsim_ckpt_tail_offset(save_words, nd, substeps, mm_freq) + dsim_tail_words(nq, nd);
}

struct DsimLayout {
    DsimDims d;
    DsimOff o;
    std::vector<uint32_t> cblob;  // const_words words
};

namespace dsim_detail {
inline uint32_t f2u(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
}  // namespace dsim_detail

// Returns "" on success, otherwise an error description.
inline std::string dsim_build_layout(const dsim_model_desc& m, DsimLayout& out) {
    using dsim_detail::f2u;
    const int L = m.n_links, nq = m.n_q, nd = m.n_qd, C = m.n_contacts, M = m.n_muscles, W = m.n_waypoints;
    if (L <= 0 || L > 64) return "n_links must be in [1,64]";
    if (nd <= 0 || nd > 64) return "n_qd must be in [1,64]";
    if (nq <= 0 || nq > 96) return "n_q must be in [1,96]";
    if (C < 0 || M < 0 || W < 0) return "negative count";
    if (m.joint_q_start[L] != nq || m.joint_qd_start[L] != nd) return "joint_q_start/joint_qd_start sentinel mismatch";
    std::vector<int> level(L), dof_link(nd, -1);
    int D = 0;
    for (int i = 0; i < L; ++i) {
        int p = m.joint_parent[i];
        if (p >= i || p < -1) return "joint_parent must precede child";
        level[i] = p < 0 ? 0 : level[p] + 1;
        if (level[i] + 1 > D) D = level[i] + 1;
        int t = m.joint_type[i];
        int ncoord = m.joint_q_start[i + 1] - m.joint_q_start[i], ndof = m.joint_qd_start[i + 1] - m.joint_qd_start[i];
        static const int ec[5] = {1, 1, 4, 0, 7}, ed[5] = {1, 1, 3, 0, 6};
        if (t < 0 || t > 4) return "unknown joint type";
        if (ncoord != ec[t] || ndof != ed[t]) return "joint coordinate/dof count does not match its type";
        for (int d = m.joint_qd_start[i]; d < m.joint_qd_start[i + 1]; ++d) dof_link[d] = i;
        // I_m must be diag(Ic (symmetric), m*1): that is what ModelBuilder.finalize produces (util.py:340-349)
        const float* I = m.body_I_m + 36 * i;
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
                bool offblock = (r < 3) != (c < 3);
                if (offblock && I[6 * r + c] != 0.f) return "body_I_m must be block diagonal";
                if (r >= 3 && c >= 3 && r != c && I[6 * r + c] != 0.f) return "body_I_m mass block must be m*identity";
            }
        if (I[21] != I[28] || I[21] != I[35]) return "body_I_m mass block must be m*identity";
        if (I[1] != I[6] || I[2] != I[12] || I[8] != I[13]) return "body_I_m rotational block must be symmetric";
        const float* xc = m.joint_X_cm + 7 * i;
        if (xc[3] != 0.f || xc[4] != 0.f || xc[5] != 0.f || xc[6] != 1.f) return "joint_X_cm rotation must be identity";
    }
    for (int c = 0; c < C; ++c)
        if (m.contact_body[c] < 0 || m.contact_body[c] >= L) return "contact_body out of range";
    for (int w = 0; w < W; ++w)
        if (m.muscle_links[w] < 0 || m.muscle_links[w] >= L) return "muscle_links out of range";
    if (M > 0 && (m.muscle_start[0] != 0 || m.muscle_start[M] != W)) return "muscle_start must span the waypoints";

    // ---- tree tables
    std::vector<int> anc_start(L + 1, 0), anc_list, sub_start(L + 1, 0), sub_list, child_start(L + 1, 0), child_list;
    std::vector<std::vector<int>> anc(L), sub(L), child(L);
    for (int i = 0; i < L; ++i) {
        std::vector<int> chain;
        for (int j = i; j != -1; j = m.joint_parent[j]) chain.push_back(j);
        anc[i].assign(chain.rbegin(), chain.rend());
        for (int j : chain) sub[j].push_back(i);  // ascending in i because i ascends
        if (m.joint_parent[i] >= 0) child[m.joint_parent[i]].push_back(i);
    }
    auto flatten = [](const std::vector<std::vector<int>>& v, std::vector<int>& start, std::vector<int>& list) {
        for (size_t i = 0; i < v.size(); ++i) {
            start[i] = (int)list.size();
            list.insert(list.end(), v[i].begin(), v[i].end());
        }
        start[v.size()] = (int)list.size();
    };
    flatten(anc, anc_start, anc_list);
    std::vector<std::vector<int>> adof(L);
    for (int i = 0; i < L; ++i)
        for (int j : anc[i])
            for (int d = m.joint_qd_start[j]; d < m.joint_qd_start[j + 1]; ++d) adof[i].push_back(d);
    std::vector<int> adof_start(L + 1, 0), adof_list;
    flatten(adof, adof_start, adof_list);
    flatten(sub, sub_start, sub_list);
    flatten(child, child_start, child_list);
    std::vector<std::vector<int>> cb(L);
    for (int c = 0; c < C; ++c) cb[m.contact_body[c]].push_back(c);
    std::vector<int> cb_start(L + 1, 0), cb_list;
    flatten(cb, cb_start, cb_list);
    auto in_subtree = [&](int a, int b) {  // is b in subtree(a)?
        for (int j = b; j != -1; j = m.joint_parent[j])
            if (j == a) return true;
        return false;
    };
    std::vector<int> rel(nd * nd, 0);
    for (int a = 0; a < nd; ++a)
        for (int b = 0; b < nd; ++b) {
            int la = dof_link[a], lb = dof_link[b];
            if (in_subtree(la, lb)) rel[a * nd + b] = 1;
            else if (in_subtree(lb, la)) rel[a * nd + b] = 2;
        }
    // ---- muscles: active segments (consecutive waypoints on different links, sim.py:1219-1223)
    std::vector<int> seg_wp, seg_m, ms_start(M + 1, 0);
    for (int mi = 0; mi < M; ++mi) {
        ms_start[mi] = (int)seg_wp.size();
        for (int w = m.muscle_start[mi]; w < m.muscle_start[mi + 1] - 1; ++w)
            if (m.muscle_links[w] != m.muscle_links[w + 1]) {
                seg_wp.push_back(w);
                seg_m.push_back(mi);
            }
    }
    ms_start[M] = (int)seg_wp.size();
    const int NS = (int)seg_wp.size();
    std::vector<std::vector<int>> ml(L);
    for (int s = 0; s < NS; ++s) {
        ml[m.muscle_links[seg_wp[s]]].push_back(2 * s + 0);
        ml[m.muscle_links[seg_wp[s] + 1]].push_back(2 * s + 1);
    }
    std::vector<int> ml_start(L + 1, 0), ml_list;
    flatten(ml, ml_start, ml_list);
    std::vector<std::vector<int>> scb(L);
    for (int i = 0; i < L; ++i)
        for (int j : sub[i]) scb[i].insert(scb[i].end(), cb[j].begin(), cb[j].end());
    std::vector<int> scb_start(L + 1, 0), scb_list;
    flatten(scb, scb_start, scb_list);

    // ---- trunk decomposition (deep pre-order trees without muscles; multi-dof joints only at the root)
    DsimDims dd;
    memset(&dd, 0, sizeof(dd));
    for (int u = 0; u < DSIM_TRUNK_MAX; ++u) dd.trunk[u] = dd.tr_par[u] = -1;
    std::vector<int> light_list;
    {
        bool pre = true, root_only = true;
        for (int i = 0; i < L; ++i) {
            for (size_t k = 0; k < sub[i].size(); ++k) pre = pre && sub[i][k] == i + (int)k;
            for (size_t k = 0; k + 1 < scb[i].size(); ++k) pre = pre && scb[i][k + 1] == scb[i][k] + 1;
            for (size_t k = 0; k + 1 < cb[i].size(); ++k) pre = pre && cb[i][k + 1] == cb[i][k] + 1;
            if (i > 0 && m.joint_qd_start[i + 1] - m.joint_qd_start[i] > 1) root_only = false;
            if (i > 0 && m.joint_parent[i] < 0) root_only = false;   // one tree
        }
        long best = -1;
        int best_c = 0;
        if (pre && root_only && L > 10 && M == 0) {
            for (int cap = 1; cap <= DSIM_LIGHT_CAP; ++cap) {
                int nt = 0, ccap = 0;
                bool ok = true;
                for (int i = 0; i < L; ++i) {
                    if ((int)sub[i].size() > cap) {
                        ++nt;
                        if ((int)child[i].size() > DSIM_TRUNK_CH) ok = false;
                    } else if ((int)scb[i].size() > ccap) {
                        ccap = (int)scb[i].size();
                    }
                }
                if (!ok || nt == 0 || nt > DSIM_TRUNK_MAX || ccap > DSIM_LIGHT_CAP) continue;
                const int passes = (6 * (L - nt) + 63) / 64;
                const long cost = (long)passes * (cap + ccap) * 14 + (long)nt * 150;   // ~cycles: issue slots of the light pass + one LDS round trip per trunk link
                if (best < 0 || cost < best) { best = cost; best_c = cap; }
            }
        }
        if (best >= 0) {
            int nt = 0;
            for (int i = 0; i < L; ++i) {
                if ((int)sub[i].size() > best_c) {
                    const int u = nt++;
                    dd.trunk[u] = i;
                    for (int v = 0; v < u; ++v)
                        if (dd.trunk[v] == m.joint_parent[i]) dd.tr_par[u] = v;
                    dd.tr_nch[u] = (int)child[i].size();
                    for (size_t k = 0; k < child[i].size(); ++k) dd.tr_ch[DSIM_TRUNK_CH * u + k] = child[i][k];
                    dd.tr_cb0[u] = cb[i].empty() ? 0 : cb[i][0];
                    dd.tr_ncb[u] = (int)cb[i].size();
                    dd.tr_d0[u] = m.joint_qd_start[i];
                    dd.tr_nd[u] = m.joint_qd_start[i + 1] - m.joint_qd_start[i];
                } else {
                    light_list.push_back(i);
                    if ((int)scb[i].size() > dd.CCAP) dd.CCAP = (int)scb[i].size();
                }
            }
            dd.NT = nt;
            dd.NLT = L - nt;
            dd.LCAP = best_c;
        }
    }

    DsimOff o;
    memset(&o, 0, sizeof(o));
    std::vector<uint32_t>& blob = out.cblob;
    blob.clear();
    auto put_i = [&](const int* p, size_t n) {
        int off = (int)blob.size();
        for (size_t k = 0; k < n; ++k) blob.push_back((uint32_t)p[k]);
        return off;
    };
    auto put_f = [&](const float* p, size_t n) {
        int off = (int)blob.size();
        for (size_t k = 0; k < n; ++k) blob.push_back(f2u(p[k]));
        return off;
    };
    o.jtype = put_i(m.joint_type, L);
    o.parent = put_i(m.joint_parent, L);
    o.qstart = put_i(m.joint_q_start, L + 1);
    o.qdstart = put_i(m.joint_qd_start, L + 1);
    o.dof_link = put_i(dof_link.data(), nd);
    // packed per-link record (16-byte aligned so that it is two ds_read_b128) + the range fast-path check
    bool ranges = true;
    std::vector<int> linfo(8 * L, 0);
    for (int i = 0; i < L; ++i) {
        const int ns = (int)sub[i].size();
        for (int k = 0; k < ns; ++k)
            if (sub[i][k] != i + k) ranges = false;
        const int nc = (int)scb[i].size();
        for (int k = 0; k + 1 < nc; ++k)
            if (scb[i][k + 1] != scb[i][k] + 1) ranges = false;
        linfo[8 * i + 0] = m.joint_parent[i];
        linfo[8 * i + 1] = m.joint_type[i];
        linfo[8 * i + 2] = m.joint_q_start[i];
        linfo[8 * i + 3] = m.joint_qd_start[i];
        linfo[8 * i + 4] = level[i];
        linfo[8 * i + 5] = ns;
        linfo[8 * i + 6] = nc ? scb[i][0] : 0;
        linfo[8 * i + 7] = nc;
    }
    while (blob.size() % 4) blob.push_back(0);
    o.linfo = put_i(linfo.data(), linfo.size());
    o.anc_start = put_i(anc_start.data(), L + 1);
    o.anc_list = put_i(anc_list.data(), anc_list.size());
    o.adof_start = put_i(adof_start.data(), L + 1);
    o.adof_list = put_i(adof_list.data(), adof_list.size());
    o.sub_start = put_i(sub_start.data(), L + 1);
    o.sub_list = put_i(sub_list.data(), sub_list.size());
    o.child_start = put_i(child_start.data(), L + 1);
    o.child_list = put_i(child_list.data(), child_list.size());
    o.light_list = put_i(light_list.data(), light_list.size());
    o.cb_start = put_i(cb_start.data(), L + 1);
    o.cb_list = put_i(cb_list.data(), cb_list.size());
    o.scb_start = put_i(scb_start.data(), L + 1);
    o.scb_list = put_i(scb_list.data(), scb_list.size());
    o.rel = put_i(rel.data(), rel.size());
    o.cbody = put_i(m.contact_body, C);
    // Rows in `mus`: chunk e owns the DSIM_MUSCLE_CHUNK rows from row e * DSIM_MUSCLE_STRIDE on; a body's last chunk is filled up
    // with rows that nothing ever writes (the work area starts out as zeros), so that a chunk sum is 16 unconditional additions,
    // and the chunks lie 17 rows = 102 words apart: (chunk, component) lanes of a wavefront then read 64 different banks
    // (16 rows = 96 words = 32 banks apart, every other chunk met the same ones).
    std::vector<int> seg_slot(2 * NS, 0);
    std::vector<int> mc_row, mc_cnt, mb_start(L + 1, 0);
    for (int i = 0; i < L; ++i) {
        mb_start[i] = (int)mc_row.size();
        for (int r = ml_start[i]; r < ml_start[i + 1]; r += DSIM_MUSCLE_CHUNK) {
            const int e = (int)mc_row.size(), n = ml_start[i + 1] - r < DSIM_MUSCLE_CHUNK ? ml_start[i + 1] - r : DSIM_MUSCLE_CHUNK;
            mc_row.push_back(e * DSIM_MUSCLE_STRIDE);
            mc_cnt.push_back(n);
            for (int j = 0; j < n; ++j) seg_slot[ml_list[r + j]] = e * DSIM_MUSCLE_STRIDE + j;
        }
    }
    mb_start[L] = (int)mc_row.size();
    const int MK = (int)mc_row.size();
    o.ms_start = put_i(ms_start.data(), M + 1);
    o.mc_row = put_i(mc_row.data(), MK);
    o.mc_cnt = put_i(mc_cnt.data(), MK);
    o.mb_start = put_i(mb_start.data(), L + 1);
    {
        while (blob.size() % 4) blob.push_back(0);
        std::vector<int> rec(8 * (size_t)NS, 0);
        for (int sgm = 0; sgm < NS; ++sgm) {
            const int w = seg_wp[sgm];
            rec[8 * sgm + 0] = 7 * m.muscle_links[w];
            rec[8 * sgm + 1] = 7 * m.muscle_links[w + 1];
            rec[8 * sgm + 2] = 3 * w;
            rec[8 * sgm + 3] = seg_m[sgm];
            rec[8 * sgm + 4] = 6 * seg_slot[2 * sgm];
            rec[8 * sgm + 5] = 6 * seg_slot[2 * sgm + 1];
        }
        o.seg_rec = put_i(rec.data(), rec.size());
    }
    o.xpj = put_f(m.joint_X_pj, 7 * L);
    std::vector<float> com(3 * L), ic6(6 * L), mass(L);
    for (int i = 0; i < L; ++i) {
        for (int k = 0; k < 3; ++k) com[3 * i + k] = m.joint_X_cm[7 * i + k];
        const float* I = m.body_I_m + 36 * i;
        ic6[6 * i + 0] = I[0]; ic6[6 * i + 1] = I[1]; ic6[6 * i + 2] = I[2];
        ic6[6 * i + 3] = I[7]; ic6[6 * i + 4] = I[8]; ic6[6 * i + 5] = I[14];
        mass[i] = I[21];
    }
    o.com = put_f(com.data(), com.size());
    o.axis = put_f(m.joint_axis, 3 * L);
    o.ic6 = put_f(ic6.data(), ic6.size());
    o.mass = put_f(mass.data(), L);
    o.tke = put_f(m.joint_target_ke, L);
    o.tkd = put_f(m.joint_target_kd, L);
    o.lke = put_f(m.joint_limit_ke, L);
    o.lkd = put_f(m.joint_limit_kd, L);
    o.target = put_f(m.joint_target, nq);
    o.lower = put_f(m.joint_limit_lower, nq);
    o.upper = put_f(m.joint_limit_upper, nq);
    o.arm = put_f(m.joint_armature, nd);
    o.cpoint = put_f(m.contact_point, 3 * C);
    o.cdist = put_f(m.contact_dist, C);
    o.cmat = put_f(m.contact_material, 4 * C);
    o.grav = put_f(m.gravity, 3);
    o.mpoints = put_f(m.muscle_points, 3 * W);
    while (blob.size() % 4) blob.push_back(0);
    o.const_words = (int)blob.size();

    int cur = o.const_words;
    auto take = [&](int n) {
        int off = cur;
        cur += (n + 3) & ~3;
        return off;
    };
    // "saved block": everything the adjoint of a substep reads from its forward pass, contiguous so that the
    // forward launch can stream it to HBM with one linear copy per substep and the adjoint launch can read it back
    // instead of recomputing it (the checkpoint row of a substep IS this block; it starts with q, qd)
    o.q = take(nq); o.qd = take(nd);
    o.xsc = take(7 * L); o.S = take(6 * nd);
    o.v = take(6 * L); o.a = take(6 * L); o.i10 = take(10 * L);
    o.ftot = take(6 * L); o.qdd = take(nd);
    o.qil = (((nd + 3) & ~3) - nd) >= 1 ? o.qdd + nd : -1;
    o.save_words = cur - o.q;
    o.act = take(nd); o.mact = take(M);
    o.ua = take(M > nd ? M : nd); o.obs = take(16 + nq + nd + (M > nd ? M : nd));
    o.f = take(6 * L); o.cw = take(6 * C); o.cwb = take(6 * L); o.tau = take(nd);
    o.ic10 = take(10 * L); o.F = take(6 * nd); o.hinv = take(nd * nd); o.prow = take(nd); o.pcol = take(nd);
    o.mus = take(6 * MK * DSIM_MUSCLE_STRIDE + NS);  // the chunks' wrench rows of 6 floats (2 NS of them written, sorted by body: seg_slot), + (adjoint) NS activation cotangents
    o.mpart = take(6 * MK);
    o.epf = take(4);
    o.fwd_words = cur;
    o.aq = take(nq); o.aqd = take(nd);
    o.aqn = o.aq; o.aqdn = o.aqd;  // integrate^T turns the output cotangents into the input cotangents in place (per-link lanes)
    o.aact = take(nd); o.amact = take(M);
    o.aqdd = take(nd); o.atau = take(nd + 1);  /* + the "zero dof": a word that stays 0 (padding of the register-resident ancestor-dof lists) */ o.aS = take(6 * nd); o.af = take(6 * L);
    o.acx = take(12 * C);   // per contact: cotangent wrench of the body's pose (6) + cotangent of the body's twist (6)
    o.av = take(6 * L); o.aa = take(6 * L); o.aatot = take(6 * L); o.avtot = take(6 * L); o.avj = take(6 * L);
    o.aw = take(6 * L);     // pose cotangent of a link as a world-frame wrench (torque about the origin, force)
    o.awp = take(6 * L);    // the same for the link's joint frame X_sj (what its motion subspace is attached to)
    o.azs = take(6 * L);    // subtree sums of aw + awp
    o.ai10m = take(10 * L); o.aic10 = take(10 * (nd > L ? nd : L)); o.aH = take(nd * nd);
    o.gua = take(M > nd ? M : nd); o.agx = take(12 * L);  // per-body sums (models with muscles): pose wrench of muscles + contacts (6), twist cotangent of contacts (6)
    // spare words behind the last array: the bounded range sums of the specialised kernels load a fixed number of entries
    // from a range's first element without clamping (dsim_range_sum_b); the longest overrun is < 32 rows of 12 floats
    cur += DSIM_TAIL_PAD;
    o.total_words = cur;

    out.o = o;
    dd.MK = MK;
    for (int p = 0; p < DSIM_PMASK_N; ++p) dd.pident |= 1 << p;
    for (int i = 0; i < L; ++i) {
        const float* x = m.joint_X_pj + 7 * i;
        const bool ident = x[3] == 0.f && x[4] == 0.f && x[5] == 0.f && x[6] == 1.f;
        const int p = (int)anc[i].size() - 1;
        if (!ident) dd.pident &= ~(1 << (p < DSIM_PMASK_N ? p : DSIM_PMASK_N - 1));
        if (!ident && p >= DSIM_PMASK_N) dd.pident = 0;
    }
    dd.L = L; dd.nq = nq; dd.nd = nd; dd.C = C; dd.M = M; dd.W = W; dd.NS = NS; dd.D = D;
    dd.flags = ranges ? DSIM_F_RANGES : 0;
    {
        bool jw = L >= 1;
        for (int i = 0; i < L; ++i) {
            const int t = m.joint_type[i], n = m.joint_qd_start[i + 1] - m.joint_qd_start[i];
            const bool hinge = (t == DSIM_JOINT_PRISMATIC || t == DSIM_JOINT_REVOLUTE) && n == 1;
            jw = jw && (hinge || (i == 0 && t == DSIM_JOINT_FREE && n == 6));
        }
        dd.JW_OK = jw ? 1 : 0;
        dd.JW_FREE_ROOT = (jw && m.joint_type[0] == DSIM_JOINT_FREE) ? 1 : 0;
    }
    for (int i = 0; i < L; ++i) {
        int sd = 0;
        for (int j : sub[i])
            if (j != i) sd += m.joint_qd_start[j + 1] - m.joint_qd_start[j];
        if (sd > dd.SDMAX) dd.SDMAX = sd;
        if ((int)adof[i].size() > dd.ADMAX) dd.ADMAX = (int)adof[i].size();
    }
    for (int i = 0; i < L; ++i)
        if ((int)cb[i].size() > dd.CBMAX) dd.CBMAX = (int)cb[i].size();
    if (L <= 32 && ranges) {
        // steps of the row-tree sums: levels deepest first, distances ascending within a level
        int n = 0;
        bool ok = true;
        auto push = [&](int kind, int lv_or_parent, int d_or_child) {
            if (n == DSIM_RT_MAX) { ok = false; return; }
            dd.rt_kind[n] = kind; dd.rt_lvl[n] = lv_or_parent; dd.rt_d[n] = d_or_child;
            ++n;
        };
        for (int lv = D - 1; lv >= 1 && ok; --lv)
            for (int dist = 1; dist < 32 && ok; ++dist) {
                bool any = false, in_row = dist <= 15;
                for (int i = 0; i < L; ++i)
                    if (level[i] == lv && i - m.joint_parent[i] == dist) {
                        any = true;
                        in_row = in_row && (i >> 4) == ((i - dist) >> 4);
                    }
                if (!any) continue;
                if (dist == 1) push(DSIM_RT_WAVE1, lv, 1);
                else if (in_row) push(DSIM_RT_ROW, lv, dist);
                else
                    for (int i = 0; i < L && ok; ++i)
                        if (level[i] == lv && i - m.joint_parent[i] == dist) push(DSIM_RT_FAR, m.joint_parent[i], i);
            }
        for (int i = 1; i < L; ++i) ok = ok && m.joint_parent[i] >= 0;   // one tree
        dd.RT_N = (ok && L > 1) ? n : 0;
        if (!dd.RT_N)
            for (int k = 0; k < DSIM_RT_MAX; ++k) dd.rt_lvl[k] = dd.rt_d[k] = dd.rt_kind[k] = 0;
        if (dd.RT_N && nd <= 16) {
            bool sh = true;
            const int ndr = m.joint_qd_start[1] - m.joint_qd_start[0];
            int dsh = 0;
            for (int i = 1; i < L; ++i) {
                sh = sh && (m.joint_qd_start[i + 1] - m.joint_qd_start[i] == 1);
                if (i == 1) dsh = m.joint_qd_start[i] - i;
                sh = sh && (m.joint_qd_start[i] - i == dsh);
            }
            if (sh && dsh >= -15 && dsh <= 15) {
                dd.DSH_OK = 1;
                dd.DSH = dsh;
                dd.ND_ROOT = ndr;
            }
        }
    }
    for (int i = 0; i < L; ++i) {
        dd.tmask |= DSIM_TM(m.joint_type[i]);
        for (size_t p = 0; p < anc[i].size() && p < DSIM_PMASK_N; ++p) dd.pmask[p] |= DSIM_TM(m.joint_type[anc[i][p]]);
    }
    out.d = dd;
    return "";
}
