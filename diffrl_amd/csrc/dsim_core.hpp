// dsim_core.hpp -- per-environment forward substep and its hand-derived adjoint, written as
// SPMD "phases" over the 64 lanes of ONE wavefront (one environment per workgroup, all state in LDS).
//
// Every phase is `ex.run([&](int lane){...})`: on the GPU `run` executes the body for its lane and then orders the
// workgroup's LDS traffic (one wavefront: a compiler fence -- the LDS executes a wave's operations in issue order; several
// wavefronts: a barrier; see dsim_hip.hip).  A phase never reads LDS words that another lane writes in the same phase
// except behind an explicit `ex.lds_fence()`, so cross-lane traffic goes LDS-write -> boundary -> LDS-read, or through the
// wavefront's cross-lane primitives (`ex.shfl`, `ex.bcast`).  `ex.fork_join(fm, fh)` is a phase of two blocks that touch
// disjoint LDS words: a single wavefront runs them one after the other, the helper-wave kernels run fh on a second
// wavefront.  What a lane keeps in registers across phases it gets from the executor (topology records, accumulators of
// the mass-matrix cotangent, early-load and prefetch registers).
// (tests/emu/ re-uses this file with an executor that runs the lanes one after another on the host
// to unit-test the phase logic without a GPU; that harness is test-only and not part of the library.)
//
// What is computed (reference: dflex/dflex/sim.py, one call of SemiImplicitIntegrator._simulate,
// sim.py:2316-2601), restructured for a wavefront instead of "one thread per articulation":
//   kinematics  level-synchronous over the tree: X_sj, X_sc, COM, motion subspace S, joint twist,
//               v, a (sim.py:1638-1678, 1323-1387, 1716-1763), world inertia in 10-parameter rigid
//               form instead of the dense T^T I T product (sim.py:1117-1134), body force (1765-1786)
//   contacts    one lane per contact point, deterministic per-body gather (sim.py:1137-1206)
//   muscles     one lane per active segment (sim.py:1209-1265)
//   tau         flat subtree sums instead of the serial leaf-to-root loop (sim.py:1792-1842)
//   mass matrix composite-rigid-body form of H = J^T M J (sim.py:2475-2545; spatial.h:691-815 and the
//               two dense gemms are never formed), + armature, explicit inverse by Gauss-Jordan
//               instead of Cholesky + two substitutions per substep (matnn.h:140-230)
//   integrate   sim.py:1505-1636
// and the reverse sweep with the reference's adjoint conventions.
#pragma once
#include <type_traits>

#include "dsim_layout.hpp"
#include "dsim_math.hpp"

// DSIM_OPAQUE(x): the kernels define it as an empty asm that "modifies" x.  Used on the per-lane joint type / chain length
// read from the register-resident topology records at the top of a phase: otherwise every lane predicate derived from
// them (~50 exec masks) is hoisted out of the substep loop as a loop invariant and kept in an SGPR pair -- far more than
// the 102 SGPRs there are, so they were spilled to VGPR lanes and read back with v_readlane at every use (measured:
// 105-178 SGPR spills per kernel).  Recomputing a mask where it is used is one v_cmp.
#ifndef DSIM_OPAQUE
#define DSIM_OPAQUE(x) (void)0
#endif
#define DSIM_NL 64  // lanes of one wavefront; the lanes of an environment's workgroup are Exec::NL = 64 * waves

typedef int __attribute__((may_alias)) dsim_int_a;

// O / D are either the runtime structs of dsim_layout.hpp (generic kernels: offsets and sizes live in SGPRs)
// or generated all-constexpr structs (dsim_static_layouts.hpp: per-model specialised kernels in which every
// LDS offset is an instruction immediate and every size a compile-time loop bound).
// LEAN: checkpoint mode (include/dsim.h: DSIM_CKPT_LEAN) -- a compile-time property of the kernels, so that the full-mode
// adjoint does not carry the forward phases the lean mode recomputes (they cost it registers: measured 2 -> 1 waves/SIMD).
template <class O, class D, bool LEAN_ = false> struct DsimCtxT {
    static constexpr bool LEAN = LEAN_;
    float* s;  // LDS image base
    // ... of the model constants, [0, const_words) of the image.  The same pointer, except where two environments of one
    // workgroup share ONE copy of the constants (dsim_hip.hip: DSIM_MODE_PAIR): then `s` is what makes the work-array offsets
    // land in this environment's own work area, and differs between the lanes of the two halves of the wave.
    const float* k;
    O o;
    D d;
    float h;   // substep length
};
typedef DsimCtxT<DsimOff, DsimDims> DsimCtx;
// words of one substep's checkpoint row: the saved block (everything the adjoint reads) or, in the lean mode, only (q, qd)
template <class Ctx> DSIM_FN int dsim_row(const Ctx& c) { return Ctx::LEAN ? c.o.xsc - c.o.q : c.o.save_words; }

#define CI(name) (reinterpret_cast<const dsim_int_a*>(c.k) + c.o.name)
#define CF(name) (c.k + c.o.name)
#define WF(name) (c.s + c.o.name)

// Small reductions out of LDS.  One wavefront per environment means nothing hides LDS latency except ILP,
// so the loops are unrolled by four with the loads grouped: (index loads) -> (data loads) -> adds, i.e. two
// LDS round trips per four terms instead of two per term.  Summation order is unchanged.
DSIM_FN float dsim_gather_sum(const float* data, int stride, int comp, const dsim_int_a* list, int b0, int b1, float acc) {
    int e = b0;
    for (; e + 4 <= b1; e += 4) {
        const int i0 = list[e], i1 = list[e + 1], i2 = list[e + 2], i3 = list[e + 3];
        const float x0 = data[stride * i0 + comp], x1 = data[stride * i1 + comp], x2 = data[stride * i2 + comp],
                    x3 = data[stride * i3 + comp];
        acc = (((acc + x0) + x1) + x2) + x3;
    }
    for (; e < b1; ++e) acc += data[stride * list[e] + comp];
    return acc;
}
// sum_{j in [first, first+count)} data[stride*j + comp]: subtree / contact sums when the numbering is pre-order
DSIM_FN float dsim_range_sum(const float* data, int stride, int comp, int first, int count, float acc) {
    const float* p = data + stride * first + comp;
    int e = 0;
    for (; e + 8 <= count; e += 8) {  // long ranges (muscle rows of a body): eight loads per LDS round trip
        const float x0 = p[stride * e], x1 = p[stride * (e + 1)], x2 = p[stride * (e + 2)], x3 = p[stride * (e + 3)],
                    x4 = p[stride * (e + 4)], x5 = p[stride * (e + 5)], x6 = p[stride * (e + 6)], x7 = p[stride * (e + 7)];
        acc = (((((((acc + x0) + x1) + x2) + x3) + x4) + x5) + x6) + x7;
    }
    for (; e + 4 <= count; e += 4) {
        const float x0 = p[stride * e], x1 = p[stride * (e + 1)], x2 = p[stride * (e + 2)], x3 = p[stride * (e + 3)];
        acc = (((acc + x0) + x1) + x2) + x3;
    }
    for (; e < count; ++e) acc += p[stride * e];
    return acc;
}
// Per-model specialised kernels (all-constexpr layout structs, which are empty classes) know an upper bound of every
// list length at compile time.  One wavefront per SIMD means every DEPENDENT LDS round trip costs its full latency, and
// a loop with a run-time trip count is a chain of them; with a bound B the loads of a whole list are issued together
// (entries past the end re-read entry 0 and are discarded), then summed in the usual order: one round trip, not count/4.
template <class Ctx> struct DsimIsStatic {
    static constexpr bool value = std::is_empty<decltype(Ctx::d)>::value;
};
// The loads are unconditional and unclamped -- entry e is at a compile-time offset from the range's first element, so its
// address costs no instruction -- and may therefore run past the end of the range (and of the array: the LDS image ends
// with DSIM_TAIL_PAD spare words, dsim_layout.hpp); what they fetch there is discarded by the select.
template <int B>
DSIM_FN float dsim_range_sum_b(const float* data, int stride, int comp, int first, int count, float acc) {
    const float* p = data + comp + stride * first;
    float x[B];
#pragma unroll
    for (int e = 0; e < B; ++e) x[e] = p[stride * e];
#pragma unroll
    for (int e = 0; e < B; ++e) acc += (e < count) ? x[e] : 0.f;
    if (count > B) acc = dsim_range_sum(data, stride, comp, first + B, count - B, acc);
    return acc;
}
// ... and for ranges that are filled up to B entries with zeros (the chunks of muscle rows, dsim_layout.hpp: seg_slot): B
// unconditional additions; x + 0 = x, so the result is that of dsim_range_sum over the entries that exist (up to the sign of a zero)
// (G: entries requested per round trip -- B at once, or in groups where the registers are short; same additions in the same order)
template <int B, int G = B> DSIM_FN float dsim_range_sum_all(const float* data, int stride, int comp, int first) {
    static_assert(B % G == 0, "groups tile the range");
    const float* p = data + comp + stride * first;
    float acc = 0.f;
#pragma unroll
    for (int g = 0; g < B; g += G) {
        float x[G];
#pragma unroll
        for (int e = 0; e < G; ++e) x[e] = p[stride * (g + e)];
        if (G < B) asm volatile("" ::: "memory");   // (keeps the groups apart: the next group's loads are not hoisted over these additions)
#pragma unroll
        for (int e = 0; e < G; ++e) acc += x[e];
    }
    return acc;
}
// The same with a per-entry weight m[e] in {1.0f, 0.0f} instead of the comparison + select: x * 1 is exact and
// acc + 0 leaves acc alone, so the result is that of dsim_range_sum_b (up to the sign of a zero) for one fused
// multiply-add per entry instead of three instructions.  The weights are per-lane constants of the launch kept in
// registers (DsimTopoRegs::lmask / cmask).  Entries past the end of the range must be FINITE for this to work: they are
// other rows of the same array or, past its end, whatever array follows in the LDS image, which is why the kernels
// clear the work area of the image once per launch (dsim_hip.hip: start_env).
template <int B>
DSIM_FN float dsim_range_sum_m(const float* data, int stride, int comp, int first, const float* m, float acc) {
    const float* p = data + comp + stride * first;
    float x[B];
#pragma unroll
    for (int e = 0; e < B; ++e) x[e] = p[stride * e];
#pragma unroll
    for (int e = 0; e < B; ++e) acc = __builtin_fmaf(x[e], m[e], acc);
    return acc;
}
// Bounds: whole lists for small trees (each lane makes one pass over its items); 8 for larger models, where the lanes
// loop over several items and loading a long mostly-unused tail per item costs more issue slots than it saves latency.
#ifndef DSIM_GENERIC_BATCH
#define DSIM_GENERIC_BATCH 8   // entries the generic kernels' range sums request per round trip (the LDS image's spare tail covers the overrun)
#endif
template <class D> constexpr int dsim_cap_links() { return D::L <= 10 ? D::L : 8; }
template <class D> constexpr int dsim_cap_subtree_contacts() { return D::C > 32 ? 8 : (D::C > 0 ? D::C : 1); }
template <class D> constexpr int dsim_cap_body_contacts() { return D::C >= 8 ? 8 : (D::C > 0 ? D::C : 1); }
struct __attribute__((aligned(16), may_alias)) dsim_i4 {
    int x, y, z, w;
};
struct __attribute__((aligned(16), may_alias)) dsim_f4 {
    float x, y, z, w;
};

// packed per-link record (dsim_layout.hpp: linfo): two 16-byte LDS reads instead of a chain of dependent 4-byte reads
struct DsimLinkInfo {
    int parent, type, cs, ds, level, nsub, c0, nc;
};
template <class Ctx> DSIM_FN DsimLinkInfo dsim_link_info(const Ctx& c, int i) {
    const dsim_i4* p = reinterpret_cast<const dsim_i4*>(reinterpret_cast<const dsim_int_a*>(c.k) + c.o.linfo + 8 * i);
    const dsim_i4 a = p[0], b = p[1];
    return DsimLinkInfo{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}
// subtree sum of a per-link array: contiguous range when the model is numbered in pre-order, CSR list otherwise
template <class Ctx> DSIM_FN float dsim_subtree_sum(const Ctx& c, const float* data, int stride, int comp, int i, int n_known = -1) {
    if (c.d.flags & DSIM_F_RANGES) {
        const int n = n_known >= 0 ? n_known : reinterpret_cast<const dsim_int_a*>(c.k)[c.o.linfo + 8 * i + 5];
        if constexpr (DsimIsStatic<Ctx>::value)
            return dsim_range_sum_b<dsim_cap_links<decltype(c.d)>()>(data, stride, comp, i, n, 0.f);
        else   // (generic kernels, round 5: the first eight entries in ONE round trip, whatever the count; the rest in the loop)
            return dsim_range_sum_b<DSIM_GENERIC_BATCH>(data, stride, comp, i, n, 0.f);
    }
    return dsim_gather_sum(data, stride, comp, CI(sub_list), CI(sub_start)[i], CI(sub_start)[i + 1], 0.f);
}

DSIM_FN float dsim_dot_n(const float* a, const float* b, int n) {
    float acc = 0.f;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        const float a0 = a[j], a1 = a[j + 1], a2 = a[j + 2], a3 = a[j + 3];
        const float b0 = b[j], b1 = b[j + 1], b2 = b[j + 2], b3 = b[j + 3];
        acc += a0 * b0;
        acc += a1 * b1;
        acc += a2 * b2;
        acc += a3 * b3;
    }
    for (; j < n; ++j) acc += a[j] * b[j];
    return acc;
}

// compile-time loop with a constexpr index (per-position joint-type masks of the specialised kernels)
template <int I, int N, class F> DSIM_FN void dsim_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dsim_static_for<I + 1, N>(f);
    }
}

// ================================================================================================
// forward
// ================================================================================================

// FK + motion subspace + velocities + world inertia + body force (one lane per link, see dsim_fwd_kinematics).
// constant parts of the work arrays (written once per launch): the motion subspace of a free joint is the identity
template <class Ctx, class Exec> DSIM_FN void dsim_init_static(const Ctx& c, Exec& ex) {
    ex.run([&](int lane) {
        for (int i = lane; i < c.d.L; i += Exec::NL) {
            const DsimLinkInfo li = dsim_link_info(c, i);
            if (li.type == DSIM_JOINT_FREE)
                for (int k = 0; k < 6; ++k)
                    for (int r = 0; r < 6; ++r) WF(S)[6 * (li.ds + k) + r] = (k == r) ? 1.f : 0.f;
        }
    });
}

// Chain records of the flat forward kinematics kept in REGISTERS (specialised kernels, trees of depth <= DSIM_CHAIN_MAX):
// the ancestors of a lane's link never change, so (link, type, q start, qd start) of every chain position is read from the
// LDS tables once per launch (dsim_topo_init) instead of as an index -> record -> data chain of three dependent LDS round
// trips per position in every substep.
#define DSIM_CHAIN_MAX 10
#define DSIM_SCAN_ROUNDS_MAX 4   // log-depth kinematics: trees of up to 16 levels
#ifndef DSIM_SCAN_MIN_DEPTH
#define DSIM_SCAN_MIN_DEPTH 5    // ... used from this many levels on (-DDSIM_SCAN_MIN_DEPTH=99 builds the A/B variant without it)
#endif
#define DSIM_TR_PASSES 2   // passes of the light items of a trunk-decomposed model over one wavefront
// Per-lane topology records kept in REGISTERS by the specialised kernels (the executor owns one per lane).  A lane plays
// the same roles in every substep -- link `lane`, dof `lane`, contact `lane` (forward) or `63 - lane` (adjoint) -- and the
// index records of those roles never change, so they are read from the LDS tables once per launch (dsim_topo_init)
// instead of being the first of two or three DEPENDENT LDS round trips of a phase in every substep.  Fields a kernel
// does not use cost nothing (dead registers).
#define DSIM_GX_PASSES 6   // 12 L <= 384 items over 64 lanes
#define DSIM_GX_CAP 12     // most contacts on one body the register form handles (DsimDims::CBMAX; more: the row-tree form is off)
struct DsimTopoRegs {
    int chain[4 * DSIM_CHAIN_MAX + 1];       // ancestors of link `lane`, root first: (link, type, q start, qd start); [last] = length
    int own_type, own_cs, own_ds, own_nd;    // joint of link `lane`
    int own_parent, own_level, own_ch0, own_ch1;  // its parent, tree level and child-list range
    int dof_link, dof_type, dof_cs, dof_ds;  // joint that dof `lane` belongs to
    int cbody_f, cbody_b;                    // body of contact `lane` / of contact `63 - lane`
    int six_n, six_c0, six_nc;               // link `lane / 6` (the (link, component) phases): subtree size, subtree contact range
    float lmask[10], cmask[32];              // ... and the 1 / 0 weights of the entries of its two range sums (dsim_range_sum_m)
    // trunk decomposition (DsimTrunk): the light (link, component) item of this lane in pass p -- row 6 i + k (-1: none), k,
    // subtree size, subtree contact range; for the ancestor sums: row of the first light ancestor-or-self, row of the nearest
    // trunk ancestor, bit e set <=> link top + e is an ancestor-or-self, first own dof (-1: none)
    int tl_row[DSIM_TR_PASSES], tl_k[DSIM_TR_PASSES], tl_n[DSIM_TR_PASSES], tl_c0[DSIM_TR_PASSES], tl_nc[DSIM_TR_PASSES];
    int ta_top[DSIM_TR_PASSES], ta_tp[DSIM_TR_PASSES], ta_m[DSIM_TR_PASSES], tu_d[DSIM_TR_PASSES];
    float tw_l[DSIM_TR_PASSES][DSIM_LIGHT_CAP], tw_c[DSIM_TR_PASSES][DSIM_LIGHT_CAP];   // 1 / 0 weights of the light sums' entries (adjoint kernels)
    int adof[16], adof_n;                    // dofs of the ancestors-or-self of link `(63 - lane) / 6` (adjoint of tau)
    int jmp[DSIM_SCAN_ROUNDS_MAX];           // lane of the ancestor of link `lane` at distance 2^r (none: the last lane): DsimScanFk
    float rt_w[DSIM_RT_MAX];                 // row-tree sums: 1 if link `lane` has a child of step s's level at lane + rt_d[s], else 0
    // per-body gather of the contact cotangent rows (row-tree body level): item `lane + 64 p` = (link, one of 12 components);
    // first word of the body's first contact row for that component (an in-range word where the body has none), contact count
    // (selects, not 1 / 0 weight registers: the records live in registers for the whole launch, of BOTH waves)
    int gx_row[DSIM_GX_PASSES], gx_n[DSIM_GX_PASSES];
    int gf_row[3], gf_n[3];                  // the same for the forward pass's contact wrench rows (6 components per item, 6 L <= 192 items)
};

// Row-tree form of the body-level adjoint (dsim_bwd_bodies_rowtree): specialised kernels of pre-order trees of at most 32 links
// (DsimDims::RT_N > 0).  Models without muscles: the one-wave mapping (contacts reduced per body by the side block / helper
// wavefront).  Models WITH muscles (several wavefronts per environment): the per-item and per-body phases stay on all
// wavefronts, the body level itself runs on the first wavefront alone (Exec::run_wave0), whose link lanes use the same row
// shifts.  DsimRowTreeFwd: the forward-pass counterparts (f_tot on registers, link <-> dof lane shifts), one-wave models only.
template <class Ctx, class Exec> struct DsimRowTree {
    static constexpr bool value = []() {
#ifdef DSIM_NO_ROWTREE   // (A/B builds)
        return false;
#else
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) {
            using D = decltype(Ctx::d);
            const bool tree = D::RT_N > 0 && (D::flags & DSIM_F_RANGES) != 0;
            return tree && ((Exec::WAVE_OPS && D::NS == 0 && D::CBMAX <= DSIM_GX_CAP) || (Exec::WAVE0_OPS && D::NS > 0));
        } else {
            return false;
        }
#endif
    }();
};
// The HELPER wavefront brings the next checkpoint row into LDS (row-tree models with a helper wave, full checkpoint mode): it
// reaches the top of the next substep right behind the side block of the body level -- while the main wave still works through
// the second half of that phase, which reads registers and the per-body rows only -- stores the row it prefetched one substep
// earlier, requests the one after, and meets the main wave at a workgroup barrier.  The main wave's substep then starts with that
// barrier instead of the copy (register -> LDS stores + the store -> load turnaround in front of integrate^T).
template <class Ctx, class Exec> struct DsimHelperCommit {
    static constexpr bool value = []() {
#ifdef DSIM_NO_HELPER_COMMIT   // (A/B builds)
        return false;
#else
        if constexpr (DsimRowTree<Ctx, Exec>::value) return Exec::HAS_HELPER && !Ctx::LEAN && decltype(Ctx::d)::NS == 0;
        else return false;
#endif
    }();
};
template <class Ctx, class Exec> struct DsimRowTreeFwd {
    static constexpr bool value = []() {
        if constexpr (DsimRowTree<Ctx, Exec>::value) return Exec::WAVE_OPS && decltype(Ctx::d)::NS == 0;
        else return false;
    }();
};
// Models whose one-wave kernels also exist for TWO environments per wavefront (32 lanes each; dsim_hip.hip: DSIM_MODE_PAIR):
// specialised layouts whose links sit in one 16-lane row (the row-tree shifts stay inside a half) and whose contacts and dofs
// are single-pass items of 32 lanes.
template <class D> constexpr bool dsim_pair_ok() {
    if constexpr (std::is_empty<D>::value) return D::NS == 0 && D::C <= DSIM_NL / 2 && D::nd <= DSIM_NL / 2 && D::L <= 16;
    else return false;
}
template <class Ctx> struct DsimChainRegs {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) return decltype(Ctx::d)::D <= DSIM_CHAIN_MAX;
        else return false;
    }();
};
// (link, component) items, one per lane, with pre-order (range) numbering
template <class Ctx, int NL> struct DsimSixRegs {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value)
            return 6 * decltype(Ctx::d)::L <= NL && (decltype(Ctx::d)::flags & DSIM_F_RANGES) != 0;
        else return false;
    }();
};
// ... whose range sums cover the whole subtree / contact range with one bounded pass: weights instead of selects
template <class Ctx, int NL> struct DsimSumMasks {
    static constexpr bool value = []() {
        if constexpr (DsimSixRegs<Ctx, NL>::value) return decltype(Ctx::d)::L <= 10 && decltype(Ctx::d)::C <= 32;
        else return false;
    }();
};
// Trunk decomposition of the subtree / ancestor sums (deep trees: dsim_layout.hpp DsimDims::NT): specialised kernels with one
// wavefront per environment (the steps of a sum are ordered by the wavefront's in-order LDS queue, not by barriers)
template <class Ctx, class Exec> struct DsimTrunk {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value && Exec::WAVE_OPS) {
            using D = decltype(Ctx::d);
            return D::NT > 0 && 6 * D::NLT <= DSIM_TR_PASSES * Exec::NL;
        } else {
            return false;
        }
    }();
};
template <class Ctx, int NL> struct DsimAdofRegs {  // ... and few enough dofs for the ancestor-dof list of an item to sit in registers
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value)
            return 6 * decltype(Ctx::d)::L <= NL && (decltype(Ctx::d)::flags & DSIM_F_RANGES) != 0 && decltype(Ctx::d)::nd <= 16;
        else return false;
    }();
};
// roles are "item index == lane": needs every loop of that role to be a single pass
template <class Ctx, int NL> struct DsimRoleRegs {  // link `lane` and dof `lane`
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value)
            return decltype(Ctx::d)::L <= NL && decltype(Ctx::d)::nd <= NL;
        else return false;
    }();
};
template <class Ctx, int NL> struct DsimContactRegs {  // contact `lane` / `63 - lane`
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) return decltype(Ctx::d)::C <= NL;
        else return false;
    }();
};
// Models whose per-item phases need SEVERAL wavefronts per environment (muscles: SNUHumanoid, 4 waves): the link-level work of
// a substep -- kinematics, inertias and body forces forward; the body level of the adjoint -- runs on the FIRST wavefront alone
// (a tree of <= 64 links), and the per-item work (ground contacts, muscle segments, their per-chunk sums, the checkpoint copies)
// on the OTHER wavefronts AT THE SAME TIME (Exec::fork_wave0: two different instruction streams with shared workgroup barriers
// where one needs what the other has written, Exec::mid / mid2 / side_done).  Round 4 ran these as phases one after the other,
// every wave waiting at a workgroup barrier while one of them worked: 65 % of all wave-cycles of the SNUHumanoid kernels were
// barrier waits (profiles/r04_final_snu_*).  The forward kinematics of the first wavefront are the log-depth scan
// (dsim_scan_fk_lane): the contact lanes no longer walk chains, they read published poses on the other wavefronts.
template <class Ctx, class Exec> struct DsimWideOverlap {
    static constexpr bool value = []() {
#ifdef DSIM_NO_WIDE_OVERLAP   // (A/B builds)
        return false;
#else
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) {
            using D = decltype(Ctx::d);
            return Exec::NL >= 2 * DSIM_NL && D::NS > 0 && D::L < DSIM_NL && D::nd <= DSIM_NL && D::D <= (1 << DSIM_SCAN_ROUNDS_MAX) &&
                   (D::flags & DSIM_F_RANGES) != 0;
        } else {
            return false;
        }
#endif
    }();
};
// Log-depth forward kinematics (dsim_fwd_kinematics_scan) instead of the per-lane chain walk: specialised kernels with one
// wavefront per environment, trees of DSIM_SCAN_MIN_DEPTH levels or more.  A walk costs every lane (and, SIMD-wise, the
// whole wave) `levels` chain positions of ~130 instructions; composing the links' LOCAL transforms along the ancestor chains
// by pointer jumping costs ceil(log2(levels)) rounds of ~45 (Humanoid: 10 positions -> 4 rounds), and the twists and bias
// accelerations are prefix sums over the same chains.  Contacts then read the finished poses / twists of their bodies.
template <class Ctx, int NL> struct DsimScanFk {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) {
            using D = decltype(Ctx::d);
            return NL <= DSIM_NL && D::L < NL && D::nd <= NL && D::C <= NL && D::D >= DSIM_SCAN_MIN_DEPTH &&
                   D::D <= (1 << DSIM_SCAN_ROUNDS_MAX);
        } else {
            return false;
        }
    }();
};
constexpr int dsim_scan_rounds(int levels) {
    int r = 0;
    while ((1 << r) < levels) ++r;
    return r;
}
// Shallow trees keep the chain walk (DsimChainRegs), but their contact lanes do not walk chains of their own any more (round 2:
// lane L + k re-walked the chain of contact k's body -- on the helper wavefront a second copy of the whole walk, a third of the
// forward kernel's instructions): the link lanes publish poses and twists as soon as the walk is done, hand over (Exec::mid)
// and go on with inertias and body forces while the contacts are evaluated from the published values.  Same code in both launch
// modes (helper wavefront or not): bit-identical results.
template <class Ctx, int NL> struct DsimContactsAfterWalk {
    static constexpr bool value = []() {
        if constexpr (DsimScanFk<Ctx, NL>::value) return false;
        else if constexpr (DsimChainRegs<Ctx>::value) {
            using D = decltype(Ctx::d);
            return NL <= DSIM_NL && D::C > 0 && D::C <= NL && D::L <= NL && D::nd <= NL && D::NS == 0;
        } else return false;
    }();
};
// Contacts evaluated INSIDE the kinematics phase: lane L + k walks the ancestor chain of contact k's body next to the link
// lanes (same instruction stream, so the walk costs nothing extra) and evaluates its contact from the pose and twist it
// holds in registers -- no phase boundary, no reload of X_sc / v.  Needs the chain records and L + C lanes.
template <class Ctx, int NL> struct DsimContactsInKin {
    static constexpr bool value = []() {
        if constexpr (DsimScanFk<Ctx, NL>::value || DsimContactsAfterWalk<Ctx, NL>::value) return false;   // they read the finished poses instead
        else if constexpr (DsimChainRegs<Ctx>::value) return decltype(Ctx::d)::C > 0 && decltype(Ctx::d)::L + decltype(Ctx::d)::C <= NL;
        else return false;
    }();
};
// ADJ: the launch runs adjoint phases (the forward kernels skip the records only those use)
template <bool ADJ = true, class Ctx, class Exec> DSIM_FN void dsim_topo_init(const Ctx& c, Exec& ex, int lane) {
    if constexpr (DsimTrunk<Ctx, Exec>::value) {
        using D = decltype(c.d);
        DsimTopoRegs& tp = ex.topo(lane);
#pragma unroll
        for (int p = 0; p < DSIM_TR_PASSES; ++p) {
            const int item = lane + Exec::NL * p;
            const bool on = item < 6 * D::NLT;
            const int i = CI(light_list)[on ? item / 6 : 0], k = on ? item - 6 * (item / 6) : 0;
            const DsimLinkInfo li = dsim_link_info(c, i);
            tp.tl_row[p] = on ? 6 * i + k : -1;
            tp.tl_k[p] = k;
            tp.tl_n[p] = li.nsub;
            tp.tl_c0[p] = li.c0;
            tp.tl_nc[p] = li.nc;
#pragma unroll
            for (int e = 0; e < D::LCAP; ++e) {
                tp.tw_l[p][e] = e < li.nsub ? 1.f : 0.f;
                DSIM_OPAQUE(tp.tw_l[p][e]);
            }
#pragma unroll
            for (int e = 0; e < D::CCAP; ++e) {
                tp.tw_c[p][e] = e < li.nc ? 1.f : 0.f;
                DSIM_OPAQUE(tp.tw_c[p][e]);
            }
            // ancestors-or-self of i, root first: trunk links, then light ones
            const int e0 = CI(anc_start)[i], e1 = CI(anc_start)[i + 1];
            int top = i, tpl = 0, mask = 0;
            for (int e = e0; e < e1; ++e) {
                const int a = CI(anc_list)[e];
                bool trunk = false;
                for (int u = 0; u < D::NT; ++u) trunk = trunk || a == D::trunk[u];
                if (trunk) tpl = a;
                else if (a < top) top = a;
            }
            for (int e = e0; e < e1; ++e) {
                const int a = CI(anc_list)[e];
                if (a >= top) mask |= 1 << (a - top);
            }
            tp.ta_top[p] = 6 * top + k;
            tp.ta_tp[p] = 6 * tpl + k;
            tp.ta_m[p] = mask;
            tp.tu_d[p] = (CI(qdstart)[i + 1] > CI(qdstart)[i]) ? CI(qdstart)[i] : -1;
        }
    }
    if constexpr (DsimScanFk<Ctx, Exec::NL>::value || DsimWideOverlap<Ctx, Exec>::value) {
        DsimTopoRegs& tp = ex.topo(lane);
        const int i = lane < c.d.L ? lane : 0;
        const int e0 = CI(anc_start)[i], e1 = CI(anc_start)[i + 1];   // ancestors-or-self, root first
#pragma unroll
        for (int r = 0; r < DSIM_SCAN_ROUNDS_MAX; ++r) {
            const int dist = 1 << r;
            // no such ancestor: the wave's last lane, which carries the neutral element (identity transform, zero twist)
            tp.jmp[r] = (lane < c.d.L && e1 - e0 > dist) ? CI(anc_list)[e1 - 1 - dist] : (Exec::NL < DSIM_NL ? Exec::NL : DSIM_NL) - 1;
        }
    } else if constexpr (DsimChainRegs<Ctx>::value) {
        constexpr int DEPTH = decltype(c.d)::D;
        int* ch = ex.topo(lane).chain;
        int i = lane < c.d.L ? lane : 0;
        bool walker = lane < c.d.L;
        if constexpr (DsimContactsInKin<Ctx, Exec::NL>::value) {
            if (lane >= c.d.L && lane < c.d.L + c.d.C) {
                i = CI(cbody)[lane - c.d.L];
                walker = true;
            }
        }
        const int e0 = CI(anc_start)[i], n = walker ? CI(anc_start)[i + 1] - e0 : 0;
#pragma unroll
        for (int p = 0; p < DEPTH; ++p) {
            const DsimLinkInfo li = dsim_link_info(c, CI(anc_list)[e0 + (p < n ? p : 0)]);
            ch[4 * p] = CI(anc_list)[e0 + (p < n ? p : 0)];
            ch[4 * p + 1] = li.type;
            ch[4 * p + 2] = li.cs;
            ch[4 * p + 3] = li.ds;
        }
        ch[4 * DSIM_CHAIN_MAX] = n;
    }
    if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
        DsimTopoRegs& tp = ex.topo(lane);
        const int i = lane < c.d.L ? lane : 0;
        tp.own_type = CI(jtype)[i];
        tp.own_cs = CI(qstart)[i];
        tp.own_ds = CI(qdstart)[i];
        tp.own_nd = CI(qdstart)[i + 1] - tp.own_ds;
        const DsimLinkInfo own = dsim_link_info(c, i);
        tp.own_parent = own.parent;
        tp.own_level = own.level;
        tp.own_ch0 = CI(child_start)[i];
        tp.own_ch1 = CI(child_start)[i + 1];
        const int d = lane < c.d.nd ? lane : 0;
        const int l = c.d.nd > 0 ? CI(dof_link)[d] : 0;
        tp.dof_link = l;
        tp.dof_type = CI(jtype)[l];
        tp.dof_cs = CI(qstart)[l];
        tp.dof_ds = CI(qdstart)[l];
    }
    if constexpr (DsimRowTree<Ctx, Exec>::value) {
        using D = decltype(c.d);
        DsimTopoRegs& tp = ex.topo(lane);
        // link `lane` has a child rt_d[s] lanes above iff that link's parent is `lane` (its level is then rt_lvl[s] or the step
        // belongs to another level): two independent loads per step, one LDS round trip for all of them
        int pr[D::RT_N], lv[D::RT_N];
        dsim_static_for<0, D::RT_N>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value;
            const int ch = (D::rt_kind[s_] != DSIM_RT_FAR && lane + D::rt_d[s_] < D::L) ? lane + D::rt_d[s_] : 0;
            pr[s_] = CI(linfo)[8 * ch];
            lv[s_] = CI(linfo)[8 * ch + 4];
        });
        dsim_static_for<0, D::RT_N>([&](auto ss) {
            constexpr int s_ = decltype(ss)::value;
            if constexpr (D::rt_kind[s_] == DSIM_RT_FAR)   // one edge: child lane rt_d -> parent lane rt_lvl
                tp.rt_w[s_] = lane == D::rt_lvl[s_] ? 1.f : 0.f;
            else
                tp.rt_w[s_] = (lane + D::rt_d[s_] < D::L && pr[s_] == lane && lv[s_] == D::rt_lvl[s_]) ? 1.f : 0.f;
            DSIM_OPAQUE(tp.rt_w[s_]);
        });
        constexpr bool FWD = DsimRowTreeFwd<Ctx, Exec>::value;   // (models with muscles gather per body in phases of their own)
        constexpr int GXP = (ADJ && FWD) ? (12 * D::L + Exec::NL - 1) / Exec::NL : 0, GFP = FWD ? (6 * D::L + Exec::NL - 1) / Exec::NL : 0;
        static_assert(GXP <= DSIM_GX_PASSES && GFP <= 3, "per-body gather passes");
#pragma unroll
        for (int p = 0; p < GFP; ++p) {
            const int it = lane + Exec::NL * p, li = it < 6 * D::L ? it / 6 : 0, r = it < 6 * D::L ? it - 6 * li : 0;
            const int b0 = CI(cb_start)[li], n = it < 6 * D::L ? CI(cb_start)[li + 1] - b0 : 0;
            const int first = (n > 0 && D::C > 0) ? CI(cb_list)[b0] : 0;
            tp.gf_row[p] = 6 * first + r;
            tp.gf_n[p] = n;
        }
#pragma unroll
        for (int p = 0; p < GXP; ++p) {
            const int it = lane + Exec::NL * p, li = it < 12 * D::L ? it / 12 : 0, r = it < 12 * D::L ? it - 12 * li : 0;
            const int b0 = CI(cb_start)[li], n = it < 12 * D::L ? CI(cb_start)[li + 1] - b0 : 0;
            // (pre-order ranges: the contacts of a body are consecutive, cb_list[b0] is the first; a body without contacts and
            // the lanes past the last item point at row 0 with all-zero weights)
            const int first = (n > 0 && D::C > 0) ? CI(cb_list)[b0] : 0;
            tp.gx_row[p] = 12 * first + r;
            tp.gx_n[p] = n;
        }
    }
    if constexpr (DsimContactRegs<Ctx, Exec::NL>::value) {
        DsimTopoRegs& tp = ex.topo(lane);
        const int kf = lane < c.d.C ? lane : 0, kb = (Exec::NL - 1 - lane) < c.d.C ? (Exec::NL - 1 - lane) : 0;
        tp.cbody_f = c.d.C > 0 ? CI(cbody)[kf] : 0;
        tp.cbody_b = c.d.C > 0 ? CI(cbody)[kb] : 0;
    }
    if constexpr (DsimSixRegs<Ctx, Exec::NL>::value) {
        DsimTopoRegs& tp = ex.topo(lane);
        const DsimLinkInfo li = dsim_link_info(c, lane < 6 * c.d.L ? lane / 6 : 0);
        tp.six_n = li.nsub;
        tp.six_c0 = li.c0;
        tp.six_nc = li.nc;
        if constexpr (DsimSumMasks<Ctx, Exec::NL>::value) {
#pragma unroll
            for (int e = 0; e < decltype(c.d)::L; ++e) {
                tp.lmask[e] = e < li.nsub ? 1.f : 0.f;
                DSIM_OPAQUE(tp.lmask[e]);   // a register, not a comparison the compiler re-derives at every use
            }
#pragma unroll
            for (int e = 0; e < (decltype(c.d)::C > 0 ? decltype(c.d)::C : 1); ++e) {
                tp.cmask[e] = e < li.nc ? 1.f : 0.f;
                DSIM_OPAQUE(tp.cmask[e]);
            }
        }
        if constexpr (DsimAdofRegs<Ctx, Exec::NL>::value) {
            const int it = Exec::NL - 1 - lane;
            const int j = it < 6 * c.d.L ? it / 6 : 0;
            const int e0 = CI(adof_start)[j], cnt = CI(adof_start)[j + 1] - e0;
            tp.adof_n = cnt;
#pragma unroll
            // entries past the end of the list name the "zero dof" nd: atau[nd] is a spare word that stays 0 (dsim_layout.hpp)
            for (int u = 0; u < decltype(c.d)::nd; ++u) {
                const int dof = CI(adof_list)[e0 + (u < cnt ? u : 0)];
                tp.adof[u] = u < cnt ? dof : (int)decltype(c.d)::nd;
            }
        }
    }
}

// the model's only free joint is its root, whose joint frame has an identity rotation (every floating-base model here)
template <class Ctx> struct DsimFreeRootIdent {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) {
            using D = decltype(Ctx::d);
            bool ok = (D::pident & 1) != 0 && D::D <= DSIM_PMASK_N;
            for (int p = 1; p < DSIM_PMASK_N; ++p) ok = ok && (D::pmask[p] & DSIM_TM(DSIM_JOINT_FREE)) == 0;
            return ok;
        } else {
            return false;
        }
    }();
};
// joint types that can occur at chain position P (specialised kernels: a compile-time constant, so the branches of
// types that do not occur there -- and their loads -- are not even compiled)
template <class D, int P> constexpr int dsim_pos_mask() { return P < DSIM_PMASK_N ? D::pmask[P] : D::tmask; }
// every joint at chain position P has an identity X_pj rotation (true of all joints of the MJCF / SNU models): the joint
// frame's orientation IS the parent link's, and q (x) identity is q exactly -- the product is not evaluated
template <class D, int P> constexpr bool dsim_pos_ident() { return P < DSIM_PMASK_N && ((D::pident >> P) & 1) != 0; }
// number of joint coordinates / dofs a lane must fetch for a joint whose type is in `mask`
constexpr int dsim_mask_nq(int mask) {
    return (mask & DSIM_TM(DSIM_JOINT_FREE)) ? 7 : (mask & DSIM_TM(DSIM_JOINT_BALL)) ? 4
           : (mask & (DSIM_TM(DSIM_JOINT_PRISMATIC) | DSIM_TM(DSIM_JOINT_REVOLUTE))) ? 1 : 0;
}
constexpr int dsim_mask_nd(int mask) {
    return (mask & DSIM_TM(DSIM_JOINT_FREE)) ? 6 : (mask & DSIM_TM(DSIM_JOINT_BALL)) ? 3
           : (mask & (DSIM_TM(DSIM_JOINT_PRISMATIC) | DSIM_TM(DSIM_JOINT_REVOLUTE))) ? 1 : 0;
}

// constants of ground contact k (material, body-fixed point, offset along the normal), loaded as a block
struct DsimContactConst {
    float ke, kd, kf, mu, cdist;
    v3 cp;
};
template <class Ctx> DSIM_FN DsimContactConst dsim_contact_load(const Ctx& c, int k) {
    const float* mat = CF(cmat) + 4 * k;
    DsimContactConst cc;
    cc.ke = mat[0]; cc.kd = mat[1]; cc.kf = mat[2]; cc.mu = mat[3];
    cc.cp = ld3(CF(cpoint) + 3 * k);
    cc.cdist = CF(cdist)[k];
    return cc;
}
// wrench of a ground contact on its body with pose (xp, xq) and twist vb (sim.py:1137-1206)
DSIM_FN sv6 dsim_contact_wrench(const DsimContactConst& k, v3 xp, q4 xq, sv6 vb) {
    const float ke = k.ke, kd = k.kd, kf = k.kf, mu = k.mu;
    v3 p = xp + rotate(xq, k.cp);
    p.y -= k.cdist;
    const v3 dpdt = vb.v + cross(vb.w, p);
    const float cc = p.y;
    sv6 wr = zerosv();
    if (cc < 0.0f) {
        const float vn = dpdt.y;
        const v3 vt = mk3(dpdt.x, 0.f, dpdt.z);
        const float fn = cc * ke;
        const float fd = (vn < 0.0f ? vn : 0.0f) * kd * (0.0f - cc);
        // |vt| and 1 / |vt| (dsim_inv_len: the reference's sqrt and 1 / l with their own roundings -- friction switches regime at
        // thresholds on these values; 0 at vt = 0, where the friction force is zero too)
        const float vt2 = dot(vt, vt), ilt = dsim_inv_len(vt2), lt = vt2 * ilt;
        const float a1 = kf * lt, a2 = 0.0f - mu * cc * ke;
        const float smin = a1 < a2 ? a1 : a2;
        const v3 ft = vt * (smin * ilt);
        const v3 ftot = mk3(ft.x, fn + fd, ft.z);
        wr = mksv(cross(p, ftot), ftot);
    }
    return wr;
}

// What a link's lane carries along its ancestor chain, and what it keeps of its OWN (last) position.
struct DsimFkWalk {
    v3 psp;   // pose of the previous chain position (parent link)
    q4 rsp;
    sv6 v, a;
    // own joint (valid after the last position)
    v3 pj;
    q4 rj;
    sv6 s0, s1, s2;  // motion subspace columns of the own joint (prismatic / revolute: s0; ball: s0..s2; free: identity, not stored)
};

// One chain position: joint `j` of type `type` (coordinates at cs, dofs at ds) on top of the pose in w.  MASK: joint
// types that can occur here.  Everything the position needs is LOADED FIRST (one LDS round trip; the words past a
// joint's own coordinates are fetched but never used), then it is pure register arithmetic: no stores -- the lane
// writes its results once, after the walk, so that no load of a later position has to wait behind a store.
// FIRST: the chain's root position (parent pose = identity, v = a = 0 before it): X_sj = X_pj, v = v_j, and the bias
// acceleration v x v_j of a vector with itself is exactly zero.
// inputs of one chain position (joint constants + the joint's coordinates / velocities), loaded as a block
template <int MASK> struct DsimFkIn {
    static constexpr int NQ = dsim_mask_nq(MASK), NDF = dsim_mask_nd(MASK);
    v3 ppj, axis;
    q4 rpj;
    float qv[NQ > 0 ? NQ : 1], qdv[NDF > 0 ? NDF : 1];
};
template <int MASK, class Ctx> DSIM_FN DsimFkIn<MASK> dsim_fk_load(const Ctx& c, int j, int cs, int ds) {
    DsimFkIn<MASK> in;
    const float *q = WF(q), *qd = WF(qd);
    in.ppj = ld3(CF(xpj) + 7 * j);
    in.rpj = ldq(CF(xpj) + 7 * j + 3);
    in.axis = ld3(CF(axis) + 3 * j);
#pragma unroll
    for (int k = 0; k < DsimFkIn<MASK>::NQ; ++k) in.qv[k] = q[cs + k];
#pragma unroll
    for (int k = 0; k < DsimFkIn<MASK>::NDF; ++k) in.qdv[k] = qd[ds + k];
    return in;
}
// all positions of a lane's chain, loaded up front (specialised kernels: the chain records of positions past the end of
// the chain repeat entry 0, so the loads need no predicate); one LDS round trip for the whole walk
template <class D, int P, int N> struct DsimFkPre {
    DsimFkIn<dsim_pos_mask<D, P>()> in;
    DsimFkPre<D, P + 1, N> rest;
};
template <class D, int N> struct DsimFkPre<D, N, N> {};
template <class D, int P, int N, class Ctx> DSIM_FN void dsim_fk_load_all(const Ctx& c, const int* ch, DsimFkPre<D, P, N>& pre) {
    if constexpr (P < N) {
        pre.in = dsim_fk_load<dsim_pos_mask<D, P>()>(c, ch[4 * P], ch[4 * P + 2], ch[4 * P + 3]);
        dsim_fk_load_all<D, P + 1, N>(c, ch, pre.rest);
    }
}
template <int MASK, bool FIRST, bool IDENT = false> DSIM_FN void dsim_fk_compute(DsimFkWalk& w, const DsimFkIn<MASK>& in, int type);
template <class D, int P, int N> DSIM_FN void dsim_fk_compute_all(DsimFkWalk& w, const int* ch, int n, const DsimFkPre<D, P, N>& pre) {
    if constexpr (P < N) {
        int ty = ch[4 * P + 1];
        DSIM_OPAQUE(ty);
        if (P < n) dsim_fk_compute<dsim_pos_mask<D, P>(), P == 0, dsim_pos_ident<D, P>()>(w, pre.in, ty);
        dsim_fk_compute_all<D, P + 1, N>(w, ch, n, pre.rest);
    }
}

// chunked walk of a long chain: positions [P, P + 3) are in `cur`; the following chunk is loaded first, then `cur` is evaluated
template <class D, int P, class Ctx, class Pre>
DSIM_FN void dsim_fk_walk_chunks(const Ctx& c, DsimFkWalk& w, const int* ch, int n, const Pre& cur) {
    constexpr int E = P + 3 < D::D ? P + 3 : D::D;        // end of the current chunk
    if constexpr (E < D::D) {
        constexpr int E2 = E + 3 < D::D ? E + 3 : D::D;
        DsimFkPre<D, E, E2> nxt;
        dsim_fk_load_all<D, E, E2>(c, ch, nxt);
        dsim_fk_compute_all<D, P, E>(w, ch, n, cur);
        dsim_fk_walk_chunks<D, E>(c, w, ch, n, nxt);
    } else {
        dsim_fk_compute_all<D, P, E>(w, ch, n, cur);
    }
}

template <int MASK, bool FIRST, class Ctx> DSIM_FN void dsim_fk_position(const Ctx& c, DsimFkWalk& w, int j, int type, int cs, int ds) {
    dsim_fk_compute<MASK, FIRST>(w, dsim_fk_load<MASK>(c, j, cs, ds), type);
}
template <int MASK, bool FIRST, bool IDENT> DSIM_FN void dsim_fk_compute(DsimFkWalk& w, const DsimFkIn<MASK>& in, int type) {
    // A position at which only ONE joint type can occur (compile-time mask of the specialised kernels) needs no run-time
    // type test: every lane that evaluates the position has a joint of that type.  With the test, the compiler turns the
    // short per-type block into selects on everything it writes (~20 v_cndmask per position and lane).
    constexpr bool ONE = MASK != 0x1f && (MASK & (MASK - 1)) == 0;
    const v3 ppj = in.ppj, axis = in.axis;
    const q4 rpj = in.rpj;
    const float *qv = in.qv, *qdv = in.qdv;
    v3 pj = ppj;
    q4 rj = IDENT ? mkq(0.f, 0.f, 0.f, 1.f) : rpj;
    if constexpr (!FIRST) {
        pj = rotate(w.rsp, ppj) + w.psp;
        if constexpr (IDENT) rj = w.rsp;
        else rj = qmul(w.rsp, rpj);
    }
    v3 pc = pj;
    q4 rc = rj;
    sv6 vj = zerosv();
    if constexpr ((MASK & DSIM_TM(DSIM_JOINT_PRISMATIC)) != 0) {
        if (ONE || type == DSIM_JOINT_PRISMATIC) {
            const v3 u = rotate(rj, axis);
            pc = pj + u * qv[0];
            w.s0 = mksv(zero3(), u);
            vj = w.s0 * qdv[0];
        }
    }
    if constexpr ((MASK & DSIM_TM(DSIM_JOINT_REVOLUTE)) != 0) {
        if (ONE || type == DSIM_JOINT_REVOLUTE) {
            rc = qmul(rj, quat_axis_angle(axis, qv[0]));
            const v3 u = rotate(rj, axis);
            w.s0 = mksv(u, cross(pj, u));
            vj = w.s0 * qdv[0];
        }
    }
    if constexpr ((MASK & DSIM_TM(DSIM_JOINT_BALL)) != 0) {
        if (ONE || type == DSIM_JOINT_BALL) {
            rc = qmul(rj, mkq(qv[0], qv[1], qv[2], qv[3]));
            v3 u0, u1, u2;
            rotate_basis(rj, u0, u1, u2);
            w.s0 = mksv(u0, cross(pj, u0));
            w.s1 = mksv(u1, cross(pj, u1));
            w.s2 = mksv(u2, cross(pj, u2));
            vj += w.s0 * qdv[0];
            vj += w.s1 * qdv[1];
            vj += w.s2 * qdv[2];
        }
    }
    if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
        if (ONE || type == DSIM_JOINT_FREE) {
            if constexpr (FIRST && IDENT) {   // identity joint frame at the root: rotate(1, x) = x and 1 (x) q = q exactly
                pc = mk3(qv[0], qv[1], qv[2]) + pj;
                rc = mkq(qv[3], qv[4], qv[5], qv[6]);
            } else {
                pc = rotate(rj, mk3(qv[0], qv[1], qv[2])) + pj;
                rc = qmul(rj, mkq(qv[3], qv[4], qv[5], qv[6]));
            }
            vj = mksv(mk3(qdv[0], qdv[1], qdv[2]), mk3(qdv[3], qdv[4], qdv[5]));  // S = identity (dsim_init_static)
        }
    }
    if constexpr (FIRST) {
        w.v = vj;
    } else {
        w.v = w.v + vj;
        w.a = w.a + scross(w.v, vj);
    }
    w.pj = pj;
    w.rj = rj;
    w.psp = pc;
    w.rsp = rc;
}

// checkpoint row of a substep: the saved block in LDS -> this environment's row in global memory, 16 bytes per lane and
// instruction.  Specialised kernels: the row length is a compile-time constant, so the copy is unrolled -- ALL the LDS reads
// are issued first, then the stores (the run-time loop it replaces waited for every read before its store: five LDS
// latencies in a row for the humanoid's 259 16-byte words).
typedef float __attribute__((vector_size(16), may_alias)) dsim_vec4;   // a native vector: stays in registers (an array of the
                                                                         // may_alias STRUCT dsim_f4 went to scratch memory)
// PART 0: the whole row; 1: its head (q, qd -- the words the integrator overwrites); 2: the rest (X_sc ... qdd).  The
// helper-wave kernels copy the head while the kinematics run (q, qd are stable until the integrator) and the rest beside the
// integrator (dsim_fwd_dynamics_wave), so that no copy is left on the main wave's path.
template <class Ctx, int NL, int PART = 0> DSIM_FN void dsim_ckpt_store_row(const Ctx& c, int lane, float* g_row) {
    if constexpr (DsimIsStatic<Ctx>::value) {
        using O = decltype(c.o);
        constexpr int ALL4 = (Ctx::LEAN ? O::xsc - O::q : O::save_words) / 4, HEAD4 = (O::xsc - O::q) / 4;
        constexpr int B4 = PART == 2 ? HEAD4 : 0, E4 = PART == 1 ? HEAD4 : ALL4, IT = (E4 - B4 + NL - 1) / NL;
        const dsim_vec4* src = reinterpret_cast<const dsim_vec4*>(WF(q)) + B4;
        dsim_vec4* dst = reinterpret_cast<dsim_vec4*>(g_row) + B4;
        dsim_vec4 x[IT > 0 ? IT : 1];
#pragma unroll
        for (int r = 0; r < IT; ++r) {
            const int k = lane + NL * r;
            if (k < E4 - B4) x[r] = src[k];
        }
#pragma unroll
        for (int r = 0; r < IT; ++r) {
            const int k = lane + NL * r;
            if (k < E4 - B4) dst[k] = x[r];
        }
    } else {
        const int head4 = (c.o.xsc - c.o.q) / 4, all4 = dsim_row(c) / 4;
        const int b4 = PART == 2 ? head4 : 0, e4 = PART == 1 ? head4 : all4;
        const dsim_f4* src = reinterpret_cast<const dsim_f4*>(WF(q));
        dsim_f4* dst = reinterpret_cast<dsim_f4*>(g_row);
        for (int k = b4 + lane; k < e4; k += NL) dst[k] = src[k];
    }
}

// ground contacts of this lane from the finished pose and twist of their bodies in LDS (sim.py:1137-1206)
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_contacts(const Ctx& c, Exec& ex, int lane) {
    for (int k = lane; k < c.d.C; k += Exec::NL) {
        int b;
        if constexpr (DsimContactRegs<Ctx, Exec::NL>::value) b = ex.topo(lane).cbody_f;
        else b = CI(cbody)[k];
        const v3 xp = ld3(WF(xsc) + 7 * b);
        const q4 xq = ldq(WF(xsc) + 7 * b + 3);
        const sv6 vb = ldsv(WF(v) + 6 * b);
        stsv(WF(cw) + 6 * k, dsim_contact_wrench(dsim_contact_load(c, k), xp, xq, vb));
    }
}

// Row-tree models: the contact wrenches of a body summed per body right behind their evaluation (same lanes, same wave: the side
// block of the kinematics phase), so that the dynamics phase finds ONE row per link and sums subtrees on registers
// (dsim_fwd_ftot_rowtree) instead of walking 9 + 25 rows per (link, component) lane.
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_contacts_per_body(const Ctx& c, Exec& ex, int lane) {
    if constexpr (DsimRowTreeFwd<Ctx, Exec>::value) {
        using D = decltype(c.d);
        ex.lds_fence();
        const DsimTopoRegs& tp = ex.topo(lane);
        constexpr int GFP = (6 * D::L + Exec::NL - 1) / Exec::NL, CB = D::CBMAX > 0 ? D::CBMAX : 1;
        float x[GFP][CB];
#pragma unroll
        for (int p = 0; p < GFP; ++p)
#pragma unroll
            for (int e = 0; e < CB; ++e) x[p][e] = WF(cw)[tp.gf_row[p] + 6 * e];
#pragma unroll
        for (int p = 0; p < GFP; ++p) {
            float acc = 0.f;
            int n = tp.gf_n[p];
            DSIM_OPAQUE(n);
#pragma unroll
            for (int e = 0; e < CB; ++e) acc += (e < n) ? x[p][e] : 0.f;
            const int it = lane + Exec::NL * p;
            if (it < 6 * D::L) WF(cwb)[it] = acc;
        }
    }
}

// the block of ONE lane of the log-depth kinematics (the first wavefront's lanes; see dsim_fwd_kinematics_scan below).
// Exec::mid2(): X_sc of every link is in LDS -- a hand-over point of the several-wavefront mapping only (DsimWideOverlap: muscle
// segments may start; a no-op elsewhere); Exec::mid(): v too (contacts may start).
template <class Ctx, class Exec> DSIM_FN void dsim_scan_fk_lane(const Ctx& c, Exec& ex, int lane) {
    using D = decltype(c.d);
    constexpr int L = D::L, R = dsim_scan_rounds(D::D), MASK = D::tmask;
    constexpr int NQ = dsim_mask_nq(MASK), NDF = dsim_mask_nd(MASK);
    constexpr bool IDENT = (D::pident & ((1 << D::D) - 1)) == ((1 << D::D) - 1);   // every X_pj rotation is the identity
    constexpr bool HAS_P = (MASK & DSIM_TM(DSIM_JOINT_PRISMATIC)) != 0, HAS_R = (MASK & DSIM_TM(DSIM_JOINT_REVOLUTE)) != 0,
                   HAS_B = (MASK & DSIM_TM(DSIM_JOINT_BALL)) != 0, HAS_F = (MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0;
    const DsimTopoRegs& tp = ex.topo(lane);
    const bool on = lane < L;
    const int i = on ? lane : 0;
    int type = tp.own_type;
    DSIM_OPAQUE(type);
    const int cs = tp.own_cs, ds = tp.own_ds;
    // ---- every input of the phase in one round trip: joint constants and coordinates, body constants
    const v3 ppj = ld3(CF(xpj) + 7 * i), axis = ld3(CF(axis) + 3 * i);
    const q4 rpj = ldq(CF(xpj) + 7 * i + 3);
    float qv[NQ > 0 ? NQ : 1], qdv[NDF > 0 ? NDF : 1];
#pragma unroll
    for (int k = 0; k < NQ; ++k) qv[k] = WF(q)[cs + k];
#pragma unroll
    for (int k = 0; k < NDF; ++k) qdv[k] = WF(qd)[ds + k];
    const v3 com = ld3(CF(com) + 3 * i);
    const float* icp = CF(ic6) + 6 * i;
    const float ic0 = icp[0], ic1 = icp[1], ic2 = icp[2], ic3 = icp[3], ic4 = icp[4], ic5 = icp[5];
    const float m = CF(mass)[i];
    const v3 grav = ld3(CF(grav));
    // ---- local transform
    v3 p = ppj;
    q4 r = rpj;
    if constexpr (HAS_P) {
        if (type == DSIM_JOINT_PRISMATIC) p = ppj + (IDENT ? axis * qv[0] : rotate(rpj, axis * qv[0]));
    }
    if constexpr (HAS_R) {
        if (type == DSIM_JOINT_REVOLUTE) {
            const q4 qa = quat_axis_angle(axis, qv[0]);
            r = IDENT ? qa : qmul(rpj, qa);
        }
    }
    if constexpr (HAS_B) {
        if (type == DSIM_JOINT_BALL) {
            const q4 qb = mkq(qv[0], qv[1], qv[2], qv[3]);
            r = IDENT ? qb : qmul(rpj, qb);
        }
    }
    if constexpr (HAS_F) {
        if (type == DSIM_JOINT_FREE) {
            const v3 pf = mk3(qv[0], qv[1], qv[2]);
            const q4 qf = mkq(qv[3], qv[4], qv[5], qv[6]);
            p = ppj + (IDENT ? pf : rotate(rpj, pf));
            r = IDENT ? qf : qmul(rpj, qf);
        }
    }
    // ---- poses: pointer jumping along the ancestor chains.  T_j travels from lane j by ds_bpermute (Exec::shfl): no LDS
    // store, no store -> load ordering -- a round is seven cross-lane reads and one composition
    // (a lane without an ancestor at that distance reads the wave's last lane, which holds the identity: rotate(1, x) + 0 = x
    // and 1 (x) q = q exactly, so every lane composes unconditionally -- no selects)
    if (lane >= L) {
        p = zero3();
        r = mkq(0.f, 0.f, 0.f, 1.f);
    }
    dsim_static_for<0, R>([&](auto rr) {
        const int src = tp.jmp[decltype(rr)::value];
        const v3 pa = mk3(ex.shfl(p.x, src), ex.shfl(p.y, src), ex.shfl(p.z, src));
        const q4 ra = mkq(ex.shfl(r.x, src), ex.shfl(r.y, src), ex.shfl(r.z, src), ex.shfl(r.w, src));
        p = rotate(ra, p) + pa;
        r = qmul(ra, r);
    });
    if (on) {
        st3(WF(xsc) + 7 * i, p);
        stq(WF(xsc) + 7 * i + 3, r);
    }
    ex.stamp();
    ex.mid2();   // (several wavefronts: the poses are final -- the muscle segments of the other wavefronts may start)
    // ---- motion subspace and joint twist from the link's own pose
    sv6 s0 = zerosv(), s1 = zerosv(), s2 = zerosv(), vj = zerosv();
    if constexpr (HAS_B) {
        // a ball joint's subspace is the joint frame's basis: R_sj = R_parent (x) R_pj
        const int par = tp.own_parent, psrc = par < 0 ? lane : par;
        const q4 rpar = mkq(ex.shfl(r.x, psrc), ex.shfl(r.y, psrc), ex.shfl(r.z, psrc), ex.shfl(r.w, psrc));
        if (type == DSIM_JOINT_BALL) {
            q4 rj = rpj;
            if (par >= 0) rj = IDENT ? rpar : qmul(rpar, rpj);
            v3 u0, u1, u2;
            rotate_basis(rj, u0, u1, u2);
            s0 = mksv(u0, cross(p, u0));
            s1 = mksv(u1, cross(p, u1));
            s2 = mksv(u2, cross(p, u2));
            vj = s0 * qdv[0];
            vj += s1 * qdv[1];
            vj += s2 * qdv[2];
        }
    }
    if constexpr (HAS_P) {
        if (type == DSIM_JOINT_PRISMATIC) {
            s0 = mksv(zero3(), rotate(r, axis));
            vj = s0 * qdv[0];
        }
    }
    if constexpr (HAS_R) {
        if (type == DSIM_JOINT_REVOLUTE) {
            const v3 u = rotate(r, axis);
            s0 = mksv(u, cross(p, u));
            vj = s0 * qdv[0];
        }
    }
    if constexpr (HAS_F) {
        if (type == DSIM_JOINT_FREE)   // S = identity (dsim_init_static)
            vj = mksv(mk3(qdv[0], qdv[1], qdv[2]), mk3(qdv[3], qdv[4], qdv[5]));
    }
    // ---- twists: prefix sums of v_j along the chains
    sv6 v = vj;
    if (lane >= L) v = zerosv();   // (the neutral element of the sums, see the poses)
    dsim_static_for<0, R>([&](auto rr) {
        const int src = tp.jmp[decltype(rr)::value];
        v += mksv(mk3(ex.shfl(v.w.x, src), ex.shfl(v.w.y, src), ex.shfl(v.w.z, src)),
                  mk3(ex.shfl(v.v.x, src), ex.shfl(v.v.y, src), ex.shfl(v.v.z, src)));
    });
    if (on) stsv(WF(v) + 6 * i, v);
    // ---- COM, world inertia about the origin (Theta = R Ic R^T, A = Theta + m(c.c 1 - c c^T), h = m c) and body force.
    // Several wavefronts (DsimWideOverlap): the other wavefronts are the longer side between the hand-over points from here on
    // (measured by leaving their blocks out: the muscle segments cost the launch 8 %, chunk sums + contacts 12 %, while this
    // wavefront's work behind the LAST hand-over ran alone), so the pose-only part goes HERE, in front of `mid` -- the segments
    // are still running -- and the body force behind the accelerations, in front of the next hand-over.  Same arithmetic.
    constexpr bool EARLY = DsimWideOverlap<Ctx, Exec>::value;
    v3 cm;
    inertia10 I;
    sv6 fg;
    auto inertia = [&]() __attribute__((always_inline)) {
        cm = rotate(r, com) + p;
        v3 rx, ry, rz;
        rotate_basis(r, rx, ry, rz);
        const v3 b0 = rx * ic0 + ry * ic1 + rz * ic2;
        const v3 b1 = rx * ic1 + ry * ic3 + rz * ic4;
        const v3 b2 = rx * ic2 + ry * ic4 + rz * ic5;
        I.m = m;
        I.h = cm * m;
        const float cc = dot(cm, cm);
        I.axx = b0.x * rx.x + b1.x * ry.x + b2.x * rz.x + m * (cc - cm.x * cm.x);
        I.axy = b0.x * rx.y + b1.x * ry.y + b2.x * rz.y - m * cm.x * cm.y;
        I.axz = b0.x * rx.z + b1.x * ry.z + b2.x * rz.z - m * cm.x * cm.z;
        I.ayy = b0.y * rx.y + b1.y * ry.y + b2.y * rz.y + m * (cc - cm.y * cm.y);
        I.ayz = b0.y * rx.z + b1.y * ry.z + b2.y * rz.z - m * cm.y * cm.z;
        I.azz = b0.z * rx.z + b1.z * ry.z + b2.z * rz.z + m * (cc - cm.z * cm.z);
        const v3 mg = grav * m;
        fg = mksv(cross(cm, mg), mg);
        if (on) st_i10(WF(i10) + 10 * i, I);
    };
    auto body_force = [&](const sv6& a) __attribute__((always_inline)) {
        const sv6 fb = inertia_mul(I, a) + scross_dual(v, inertia_mul(I, v));
        if (on) {
            stsv(WF(f) + 6 * i, fb - fg);
            float* S = WF(S) + 6 * ds;
            if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
                stsv(S, s0);
            } else if (type == DSIM_JOINT_BALL) {
                stsv(S, s0);
                stsv(S + 6, s1);
                stsv(S + 12, s2);
            }
        }
    };
    if constexpr (EARLY) inertia();
    ex.stamp();
    ex.mid();   // X_sc and v of every link are final: the contacts may start
    // ---- bias accelerations: c_i = v_i x v_j,i (exactly zero at the root: a vector crossed with itself), prefix sums in a
    sv6 a = zerosv();
    if (tp.own_level > 0 && on) a = scross(v, vj);
    dsim_static_for<0, R>([&](auto rr) {
        const int src = tp.jmp[decltype(rr)::value];
        a += mksv(mk3(ex.shfl(a.w.x, src), ex.shfl(a.w.y, src), ex.shfl(a.w.z, src)),
                  mk3(ex.shfl(a.v.x, src), ex.shfl(a.v.y, src), ex.shfl(a.v.z, src)));
    });
    if (on) stsv(WF(a) + 6 * i, a);
    if constexpr (EARLY) body_force(a);
    ex.stamp();
    ex.mid2();   // (several wavefronts: chunk sums and contact wrenches of the other wavefronts are complete, their per-body gather may start)
    if constexpr (!EARLY) {
        inertia();
        body_force(a);
    }
}

// Log-depth forward kinematics (DsimScanFk).  One lane per link, every lane executes every step (lanes past the last link
// compute on link 0's inputs and store nothing): the rounds exchange their operands between lanes with ds_bpermute (Exec::shfl),
// which needs uniform control flow, and the results go to LDS once -- poses, twists and bias accelerations as they become final:
//   local    T_i = X_pj o X_jc(q_i), the transform from the parent's frame to the link's (sim.py:1269-1319)
//   poses    R rounds of pointer jumping: T_i <- T_j o T_i with j the ancestor at distance 2^r; after ceil(log2(levels)) rounds
//            T_i = X_sc[i] (sim.py:1638-1678 composes the same transforms root to leaf, one link after the other)
//   S, v_j   motion subspace and joint twist from the link's own pose: a revolute / ball joint does not translate, so the joint
//            frame's origin is the link's (p_sj = p_sc), and a rotation about the axis leaves the axis where it was
//            (R_sc axis = R_sj axis); sim.py:1323-1387
//   v        prefix sums of v_j over the ancestor chains (R rounds), then the bias term c_i = v_i x v_j,i per link and
//   a        prefix sums of c (sim.py:1716-1763: v = v_parent + v_j, a = a_parent + v x v_j)
// Same terms as the chain walk, associated differently (products of transforms pairwise instead of left to right, sums
// likewise): not bit-identical to it; tests hold both to the reference's recording of the first substep (1e-5).
// Between poses + twists and the rest, Exec::mid() lets the helper wavefront start on the contacts (fork_join_mid).
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_kinematics_scan(const Ctx& c, Exec& ex, float* g_row) {
    ex.fork_join_mid([&](int lane) { dsim_scan_fk_lane(c, ex, lane); }, [&](int lane) {
        dsim_fwd_contacts(c, ex, lane);
        dsim_fwd_contacts_per_body(c, ex, lane);
        if (g_row) dsim_ckpt_store_row<Ctx, Exec::NL, 1>(c, lane, g_row);   // head of the checkpoint row: (q, qd) entering the substep
    });
}

template <int G = DSIM_MUSCLE_CHUNK, class Ctx> DSIM_FN void dsim_muscle_chunk_sums(const Ctx& c, int lane, int nl);
// where the activation cotangents of the segments start in `mus` (behind the chunks' rows)
template <class Ctx> DSIM_FN int dsim_mus_act(const Ctx& c) { return 6 * c.d.MK * DSIM_MUSCLE_STRIDE; }
// ground contacts and muscle segments of the environment on the NLR lanes `lane` of the wavefronts behind the first one, from the
// published poses and twists (forward: wrenches; DsimWideOverlap).  Contacts are dealt from the top lane down, segments from
// lane 0 up, so that a lane's second item is of the other kind where the lists allow it.
// packed record of muscle segment s (dsim_layout.hpp: seg_rec): offsets of its two links' poses, of its waypoints, its muscle, and
// of its two rows in `mus` -- two 16-byte LDS reads
struct DsimSegRec {
    int x0, x1, wp, m, r0, r1;
};
template <class Ctx> DSIM_FN DsimSegRec dsim_seg_rec(const Ctx& c, int s) {
    const dsim_i4* p = reinterpret_cast<const dsim_i4*>(reinterpret_cast<const dsim_int_a*>(c.k) + c.o.seg_rec + 8 * s);
    const dsim_i4 a = p[0], b = p[1];
    return DsimSegRec{a.x, a.y, a.z, a.w, b.x, b.y};
}
// the segment's inputs (second LDS round trip), then arithmetic and stores: kept apart so that a lane with two segments can
// request both before it evaluates either
struct DsimSegIn {
    v3 p0, p1, m0, m1;
    q4 r0, r1;
    float act;
};
// (in two parts: what does not depend on the poses -- waypoints in link coordinates, the muscle's activation -- and the poses)
template <class Ctx> DSIM_FN void dsim_seg_load_const(const Ctx& c, const DsimSegRec& g, DsimSegIn& in) {
    in.m0 = ld3(CF(mpoints) + g.wp); in.m1 = ld3(CF(mpoints) + g.wp + 3);
    in.act = WF(mact)[g.m];
}
template <class Ctx> DSIM_FN void dsim_seg_load_pose(const Ctx& c, const DsimSegRec& g, DsimSegIn& in) {
    in.p0 = ld3(WF(xsc) + g.x0); in.r0 = ldq(WF(xsc) + g.x0 + 3);
    in.p1 = ld3(WF(xsc) + g.x1); in.r1 = ldq(WF(xsc) + g.x1 + 3);
}
template <class Ctx> DSIM_FN DsimSegIn dsim_seg_load(const Ctx& c, const DsimSegRec& g) {
    DsimSegIn in;
    dsim_seg_load_pose(c, g, in);
    dsim_seg_load_const(c, g, in);
    return in;
}
template <class Ctx> DSIM_FN void dsim_fwd_muscle_segment_eval(const Ctx& c, const DsimSegRec& g, const DsimSegIn& in) {
    const v3 pos0 = in.p0 + rotate(in.r0, in.m0), pos1 = in.p1 + rotate(in.r1, in.m1);
    const v3 d = pos1 - pos0;
    const v3 f = d * (in.act * dsim_inv_len_item(dot(d, d)));   // normalize(d) * activation (zero for d = 0)
    // wrenches on the two links, signs applied, into the rows of their bodies (rows are sorted by body, so that the
    // per-body gather is a contiguous range sum without index loads)
    float* o0 = WF(mus) + g.r0;
    float* o1 = WF(mus) + g.r1;
    st3(o0, -cross(pos0, f));
    st3(o0 + 3, -f);
    st3(o1, cross(pos1, f));
    st3(o1 + 3, f);
}
template <class Ctx> DSIM_FN void dsim_fwd_muscle_segment(const Ctx& c, int s) {
    const DsimSegRec g = dsim_seg_rec(c, s);
    dsim_fwd_muscle_segment_eval(c, g, dsim_seg_load(c, g));
}
// segments lane, lane + nl, ... of the NS segments: in pairs, both records and both input sets requested before either is
// evaluated (198 segments on 192 lanes: the few lanes with two would otherwise run two dependent chains one after the other,
// and every lane of their wavefront with them)
template <class Ctx> DSIM_FN void dsim_fwd_muscle_segments(const Ctx& c, int lane, int nl) {
    for (int s = lane; s < c.d.NS; s += 2 * nl) {
        const int s2 = s + nl;
        const bool two = s2 < c.d.NS;
        const DsimSegRec ga = dsim_seg_rec(c, s), gb = dsim_seg_rec(c, two ? s2 : s);
        const DsimSegIn ia = dsim_seg_load(c, ga), ib = dsim_seg_load(c, gb);
        dsim_fwd_muscle_segment_eval(c, ga, ia);
        if (two) dsim_fwd_muscle_segment_eval(c, gb, ib);
    }
}
// Kinematics of a model with several wavefronts per environment (DsimWideOverlap): the first wavefront runs the log-depth
// kinematics of the links; the others copy the head of the checkpoint row while they wait for the poses, evaluate the muscle
// segments from the published poses (mid2) while the first wavefront goes on with motion subspaces and twists, then (mid: twists
// published, muscle rows complete) sum the muscle rows per chunk and evaluate the ground contacts while the first wavefront does
// bias accelerations, then (mid2) gather muscle and contact wrenches per body while it does inertias and body forces; behind
// side_done_w the first wavefront adds the per-body rows to its body forces and sums f_tot over the subtrees on registers.
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_kinematics_wide(const Ctx& c, Exec& ex, float* g_row) {
    constexpr int NLR = Exec::NL - DSIM_NL;
    using D = decltype(c.d);
    ex.fork_wave0([&](int lane) {
        dsim_scan_fk_lane(c, ex, lane);
        // f_tot = subtree sums of (body force + the body's muscle and contact wrenches, gathered per body by the other wavefronts
        // meanwhile): a row-tree sum on the link lanes' registers -- round 4 ran the gather and the sums as two more phases of all
        // four wavefronts
        ex.side_done_w();
        const int i = lane < D::L ? lane : 0;
        sv6 x = ldsv(WF(f) + 6 * i) + ldsv(WF(cwb) + 6 * i);   // (lanes without a link: link 0's values, which no step's weight selects)
        dsim_rowtree_sum(c, ex, ex.topo(lane), x);
        if (lane < D::L) stsv(WF(ftot) + 6 * lane, x);
        // (the joint-space forces behind it on the same wavefront's dof lanes -- no phase of their own -- measured +-0: 0.2371 ->
        // 0.2374 ms; they stay a phase of all wavefronts)
    }, [&](int lane) {
        if (g_row) dsim_ckpt_store_row<Ctx, NLR, 1>(c, lane, g_row);   // head of the checkpoint row: (q, qd) entering the substep
        // These wavefronts are the longer side between the hand-over points (measured by leaving blocks out: without the segments the
        // forward launch is 8 % shorter, without chunk sums + contacts 12 %), and each of their blocks starts with index / constant
        // loads that do not depend on what the hand-over publishes: they are requested in FRONT of it -- records, waypoints and
        // activations of the lane's segments while the poses are still being composed, the contact's constants behind the
        // segments, the body's bounds behind the contacts -- one dependent LDS round trip less per block.
        constexpr bool PRE = D::NS <= 2 * NLR && D::C <= NLR && 6 * D::MK <= NLR && 6 * D::L <= NLR;
        if constexpr (PRE) {
            const int sa = lane < D::NS ? lane : 0, sb = lane + NLR < D::NS ? lane + NLR : sa;
            const bool one = lane < D::NS, two = lane + NLR < D::NS;
            const DsimSegRec ga = dsim_seg_rec(c, sa), gb = dsim_seg_rec(c, sb);
            DsimSegIn ia, ib;
            dsim_seg_load_const(c, ga, ia);
            dsim_seg_load_const(c, gb, ib);
            ex.mid2();   // poses
            dsim_seg_load_pose(c, ga, ia);
            dsim_seg_load_pose(c, gb, ib);
            if (one) dsim_fwd_muscle_segment_eval(c, ga, ia);
            if (two) dsim_fwd_muscle_segment_eval(c, gb, ib);
            const int kc = NLR - 1 - lane, kcl = kc < D::C ? kc : 0;
            const DsimContactConst cc = dsim_contact_load(c, kcl);
            const int cb = CI(cbody)[kcl];
            const int ce = lane / 6, ck = lane - 6 * ce;
            ex.mid();    // twists (and: every muscle row is written)
            if (ce < D::MK) WF(mpart)[lane] = dsim_range_sum_all<DSIM_MUSCLE_CHUNK>(WF(mus), 6, ck, ce * DSIM_MUSCLE_STRIDE);
            if (kc < D::C) stsv(WF(cw) + 6 * kc, dsim_contact_wrench(cc, ld3(WF(xsc) + 7 * cb), ldq(WF(xsc) + 7 * cb + 3), ldsv(WF(v) + 6 * cb)));
            const int gbd = ce < D::L ? ce : 0;   // (lane = 6 * body + component, as for the chunks)
            const int m0 = CI(mb_start)[gbd], mn = CI(mb_start)[gbd + 1] - m0, c0 = CI(cb_start)[gbd], cn = CI(cb_start)[gbd + 1] - c0;
            ex.mid2();   // chunk sums and contact wrenches of all three wavefronts
            if (ce < D::L) {   // per body: its chunks' sums + its own contact wrenches
                const float acc = dsim_range_sum_b<8>(WF(mpart), 6, ck, m0, mn, 0.f);
                WF(cwb)[lane] = dsim_range_sum_b<dsim_cap_body_contacts<D>()>(WF(cw), 6, ck, c0, cn, acc);
            }
            ex.side_done_w();
        } else {
            ex.mid2();   // poses
            dsim_fwd_muscle_segments(c, lane, NLR);
            ex.mid();    // twists (and: every muscle row is written)
            dsim_muscle_chunk_sums(c, lane, NLR);
            for (int k = NLR - 1 - lane; k < c.d.C; k += NLR) {
                const int b = CI(cbody)[k];
                stsv(WF(cw) + 6 * k, dsim_contact_wrench(dsim_contact_load(c, k), ld3(WF(xsc) + 7 * b), ldq(WF(xsc) + 7 * b + 3), ldsv(WF(v) + 6 * b)));
            }
            ex.mid2();   // chunk sums and contact wrenches of all three wavefronts
            for (int it = lane; it < 6 * c.d.L; it += NLR) {   // per body: its chunks' sums + its own contact wrenches
                const int b = it / 6, k = it - 6 * b;
                WF(cwb)[it] = dsim_body_contact_sum(c, b, WF(cw), 6, k, dsim_body_chunk_sum(c, b, k, 0.f));
            }
            ex.side_done_w();
        }
    });
}

// Chain walk of a shallow tree with the contacts behind it (DsimContactsAfterWalk).  Every lane runs the whole block (lanes past
// the last link have an empty chain and store nothing), so that the hand-over is a point all lanes pass.
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_kinematics_walk_mid(const Ctx& c, Exec& ex, float* g_row) {
    using D = decltype(c.d);
    ex.fork_join_mid([&](int lane) {
        const bool on = lane < D::L;
        const int i = on ? lane : 0;
        DsimFkWalk w;
        w.psp = zero3();
        w.rsp = mkq(0.f, 0.f, 0.f, 1.f);
        w.v = zerosv();
        w.a = zerosv();
        w.s0 = w.s1 = w.s2 = zerosv();
        // body constants of the link, requested together with the chain inputs
        const v3 com = ld3(CF(com) + 3 * i);
        const float* icp = CF(ic6) + 6 * i;
        const float ic0 = icp[0], ic1 = icp[1], ic2 = icp[2], ic3 = icp[3], ic4 = icp[4], ic5 = icp[5];
        const float m = CF(mass)[i];
        const v3 grav = ld3(CF(grav));
        const int* ch = ex.topo(lane).chain;
        int n = ch[4 * DSIM_CHAIN_MAX];
        DSIM_OPAQUE(n);
        if constexpr (D::D <= 4) {
            DsimFkPre<D, 0, D::D> pre;
            dsim_fk_load_all<D, 0, D::D>(c, ch, pre);
            dsim_fk_compute_all<D, 0, D::D>(w, ch, n, pre);
        } else {
            DsimFkPre<D, 0, 3> pre;
            dsim_fk_load_all<D, 0, 3>(c, ch, pre);
            dsim_fk_walk_chunks<D, 0>(c, w, ch, n, pre);
        }
        int own_type = ex.topo(lane).own_type;
        const int own_ds = ex.topo(lane).own_ds;
        DSIM_OPAQUE(own_type);
        const v3 pc = w.psp;
        const q4 rc = w.rsp;
        if (on) {
            st3(WF(xsc) + 7 * i, pc);
            stq(WF(xsc) + 7 * i + 3, rc);
            stsv(WF(v) + 6 * i, w.v);
        }
        ex.stamp();
        ex.mid();   // poses and twists of every link are in LDS: the contacts may start
        // COM, world inertia about the origin (Theta = R Ic R^T, A = Theta + m(c.c 1 - c c^T), h = m c) and body force
        const v3 cm = rotate(rc, com) + pc;
        v3 rx, ry, rz;
        rotate_basis(rc, rx, ry, rz);
        const v3 b0 = rx * ic0 + ry * ic1 + rz * ic2;
        const v3 b1 = rx * ic1 + ry * ic3 + rz * ic4;
        const v3 b2 = rx * ic2 + ry * ic4 + rz * ic5;
        inertia10 I;
        I.m = m;
        I.h = cm * m;
        const float cc = dot(cm, cm);
        I.axx = b0.x * rx.x + b1.x * ry.x + b2.x * rz.x + m * (cc - cm.x * cm.x);
        I.axy = b0.x * rx.y + b1.x * ry.y + b2.x * rz.y - m * cm.x * cm.y;
        I.axz = b0.x * rx.z + b1.x * ry.z + b2.x * rz.z - m * cm.x * cm.z;
        I.ayy = b0.y * rx.y + b1.y * ry.y + b2.y * rz.y + m * (cc - cm.y * cm.y);
        I.ayz = b0.y * rx.z + b1.y * ry.z + b2.y * rz.z - m * cm.y * cm.z;
        I.azz = b0.z * rx.z + b1.z * ry.z + b2.z * rz.z + m * (cc - cm.z * cm.z);
        const sv6 fb = inertia_mul(I, w.a) + scross_dual(w.v, inertia_mul(I, w.v));
        const v3 mg = grav * m;
        const sv6 fg = mksv(cross(cm, mg), mg);
        if (on) {
            stsv(WF(a) + 6 * i, w.a);
            st_i10(WF(i10) + 10 * i, I);
            stsv(WF(f) + 6 * i, fb - fg);
            float* S = WF(S) + 6 * own_ds;
            if (own_type == DSIM_JOINT_PRISMATIC || own_type == DSIM_JOINT_REVOLUTE) {
                stsv(S, w.s0);
            } else if (own_type == DSIM_JOINT_BALL) {
                stsv(S, w.s0);
                stsv(S + 6, w.s1);
                stsv(S + 12, w.s2);
            }
        }
    }, [&](int lane) {
        dsim_fwd_contacts(c, ex, lane);
        dsim_fwd_contacts_per_body(c, ex, lane);
        if (g_row) dsim_ckpt_store_row<Ctx, Exec::NL, 1>(c, lane, g_row);   // head of the checkpoint row: (q, qd) entering the substep
    });
}

// g_row (one-phase dynamics only, DsimWaveDyn): the substep's checkpoint row -- its head (q, qd) is copied here, by the helper
// wavefront where there is one, the rest beside the integrator (dsim_fwd_dynamics_wave)
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_kinematics(const Ctx& c, Exec& ex, float* g_row = nullptr) {
    ex.mark(1);
    if constexpr (DsimWideOverlap<Ctx, Exec>::value) {
        dsim_fwd_kinematics_wide(c, ex, g_row);
        return;
    }
    if constexpr (DsimScanFk<Ctx, Exec::NL>::value) {
        dsim_fwd_kinematics_scan(c, ex, g_row);
        return;
    }
    if constexpr (DsimContactsAfterWalk<Ctx, Exec::NL>::value) {
        dsim_fwd_kinematics_walk_mid(c, ex, g_row);
        return;
    }
    // "Flat" forward kinematics: every link's lane walks its own ancestor chain from the root and recomputes the
    // joint transforms / twists on the way, instead of one barrier-separated phase per tree level (redundant
    // arithmetic, no per-level phase boundaries; same operations in the same order along each chain as the
    // level-synchronous form), then goes straight on to the world inertia and the body force from registers.
    // Nothing is stored before the end of the walk: LDS stores in between would serialise the next position's loads
    // behind them (the compiler cannot prove that they do not alias).
    // role 0: every walker; 1: link lanes only; 2: contact lanes only (the two blocks of the phase when there is a helper wave)
    // (always_inline: with two call sites the inliner may hesitate, and an out-of-line call of a lambda that captures the
    // phase's state by reference would go through scratch memory)
    auto walk = [&](int lane, int role) __attribute__((always_inline)) {
        constexpr bool with_contacts = DsimContactsInKin<Ctx, Exec::NL>::value;
        const int n_walkers = with_contacts ? c.d.L + c.d.C : c.d.L;
        for (int i = lane; i < n_walkers; i += Exec::NL) {
            if (role == 1 && i >= c.d.L) continue;
            if (role == 2 && i < c.d.L) continue;
            DsimFkWalk w;
            w.psp = zero3();
            w.rsp = mkq(0.f, 0.f, 0.f, 1.f);
            w.v = zerosv();
            w.a = zerosv();
            w.s0 = w.s1 = w.s2 = zerosv();
            // body constants of the link (contact lanes: of link 0, unused), requested together with the chain inputs
            const int ib = i < c.d.L ? i : 0;
            const v3 com = ld3(CF(com) + 3 * ib);
            const float* icp = CF(ic6) + 6 * ib;
            const float ic0 = icp[0], ic1 = icp[1], ic2 = icp[2], ic3 = icp[3], ic4 = icp[4], ic5 = icp[5];
            const float m = CF(mass)[ib];
            const v3 grav = ld3(CF(grav));
            DsimContactConst ccst{};
            if constexpr (with_contacts) ccst = dsim_contact_load(c, i >= c.d.L ? i - c.d.L : 0);
            int own_type, own_ds;
            if constexpr (DsimChainRegs<Ctx>::value) {
                using D = decltype(c.d);
                const int* ch = ex.topo(lane).chain;
                int n = ch[4 * DSIM_CHAIN_MAX];
                DSIM_OPAQUE(n);
                if constexpr (D::D <= 4) {
                    // short chains: every position's inputs in ONE round trip (3 positions of an Ant leg: ~45 registers)
                    DsimFkPre<D, 0, D::D> pre;
                    dsim_fk_load_all<D, 0, D::D>(c, ch, pre);
                    dsim_fk_compute_all<D, 0, D::D>(w, ch, n, pre);
                } else {
                    // long chains (Humanoid: 10 positions): chunks of three positions, the next chunk's inputs requested
                    // before the current chunk is evaluated
                    DsimFkPre<D, 0, 3> pre;
                    dsim_fk_load_all<D, 0, 3>(c, ch, pre);
                    dsim_fk_walk_chunks<D, 0>(c, w, ch, n, pre);
                }
                own_type = ex.topo(lane).own_type;
                own_ds = ex.topo(lane).own_ds;
                DSIM_OPAQUE(own_type);
            } else {
                const int e0 = CI(anc_start)[i], e1 = CI(anc_start)[i + 1];
                DsimLinkInfo li{};
                for (int e = e0; e < e1; ++e) {
                    const int j = CI(anc_list)[e];
                    li = dsim_link_info(c, j);
                    dsim_fk_position<0x1f, false>(c, w, j, li.type, li.cs, li.ds);
                }
                own_type = li.type;
                own_ds = li.ds;
            }
            if constexpr (with_contacts) {
                if (i >= c.d.L) {
                    // contact lane: the walk ended at the contact's body; its pose and twist are in registers
                    stsv(WF(cw) + 6 * (i - c.d.L), dsim_contact_wrench(ccst, w.psp, w.rsp, w.v));
                    continue;
                }
            }
            ex.stamp();
            // COM, world inertia and body force of link i from the values still in registers
            const v3 pc = w.psp;
            const q4 rc = w.rsp;
            const v3 cm = rotate(rc, com) + pc;
            // world-frame inertia about the origin: Theta = R Ic R^T, A = Theta + m(c.c 1 - c c^T), h = m c
            v3 rx, ry, rz;
            rotate_basis(rc, rx, ry, rz);
            // B = R * Ic (columns of R are rx, ry, rz)
            const v3 b0 = rx * ic0 + ry * ic1 + rz * ic2;
            const v3 b1 = rx * ic1 + ry * ic3 + rz * ic4;
            const v3 b2 = rx * ic2 + ry * ic4 + rz * ic5;
            inertia10 I;
            I.m = m;
            I.h = cm * m;
            const float cc = dot(cm, cm);
            I.axx = b0.x * rx.x + b1.x * ry.x + b2.x * rz.x + m * (cc - cm.x * cm.x);
            I.axy = b0.x * rx.y + b1.x * ry.y + b2.x * rz.y - m * cm.x * cm.y;
            I.axz = b0.x * rx.z + b1.x * ry.z + b2.x * rz.z - m * cm.x * cm.z;
            I.ayy = b0.y * rx.y + b1.y * ry.y + b2.y * rz.y + m * (cc - cm.y * cm.y);
            I.ayz = b0.y * rx.z + b1.y * ry.z + b2.y * rz.z - m * cm.y * cm.z;
            I.azz = b0.z * rx.z + b1.z * ry.z + b2.z * rz.z + m * (cc - cm.z * cm.z);
            const sv6 fb = inertia_mul(I, w.a) + scross_dual(w.v, inertia_mul(I, w.v));
            const v3 mg = grav * m;
            const sv6 fg = mksv(cross(cm, mg), mg);
            // ---- all stores of the phase
            st3(WF(xsc) + 7 * i, pc);
            stq(WF(xsc) + 7 * i + 3, rc);
            stsv(WF(v) + 6 * i, w.v);
            stsv(WF(a) + 6 * i, w.a);
            st_i10(WF(i10) + 10 * i, I);
            stsv(WF(f) + 6 * i, fb - fg);
            float* S = WF(S) + 6 * own_ds;
            if (own_type == DSIM_JOINT_PRISMATIC || own_type == DSIM_JOINT_REVOLUTE) {
                stsv(S, w.s0);
            } else if (own_type == DSIM_JOINT_BALL) {
                stsv(S, w.s0);
                stsv(S + 6, w.s1);
                stsv(S + 12, w.s2);
            }
        }
    };
    auto head = [&](int lane) __attribute__((always_inline)) {
        if (g_row) dsim_ckpt_store_row<Ctx, Exec::NL, 1>(c, lane, g_row);
    };
    if constexpr (Exec::HAS_HELPER && DsimContactsInKin<Ctx, Exec::NL>::value)
        ex.fork_join([&](int lane) { walk(lane, 1); }, [&](int lane) { walk(lane, 2); head(lane); });
    else
        ex.run([&](int lane) { walk(lane, 0); head(lane); });
}

// mpart[e][k] = sum of component k over the muscle wrench rows of chunk e (forward: wrenches; adjoint: pose cotangents)
// G: rows requested per round trip by the specialised kernels (the adjoint's several-wavefront kernel is at its register limit: two groups of 8)
template <int G, class Ctx> DSIM_FN void dsim_muscle_chunk_sums(const Ctx& c, int lane, int nl) {
    for (int it = lane; it < 6 * c.d.MK; it += nl) {
        const int e = it / 6, k = it - 6 * e;
        // (specialised kernels: all rows of the chunk in one round trip; the rows behind its last one are zeros)
        if constexpr (DsimIsStatic<Ctx>::value) WF(mpart)[it] = dsim_range_sum_all<DSIM_MUSCLE_CHUNK, G>(WF(mus), 6, k, e * DSIM_MUSCLE_STRIDE);
        else WF(mpart)[it] = dsim_range_sum(WF(mus), 6, k, CI(mc_row)[e], CI(mc_cnt)[e], 0.f);
    }
}
// sum of the chunk sums of body i, component k (<= 8 chunks per body in one round trip where the layout is compile-time)
template <class Ctx> DSIM_FN float dsim_body_chunk_sum(const Ctx& c, int i, int k, float acc) {
    const int b0 = CI(mb_start)[i], n = CI(mb_start)[i + 1] - b0;
    if constexpr (DsimIsStatic<Ctx>::value) return dsim_range_sum_b<8>(WF(mpart), 6, k, b0, n, acc);
    else return dsim_range_sum(WF(mpart), 6, k, b0, n, acc);
}
// cotangent of muscle activation m: sum over its active segments (consecutive words behind the wrench rows)
template <class Ctx> DSIM_FN float dsim_muscle_act_sum(const Ctx& c, int m) {
    const int s0 = CI(ms_start)[m], n = CI(ms_start)[m + 1] - s0;
    if constexpr (DsimIsStatic<Ctx>::value) return dsim_range_sum_b<4>(WF(mus) + dsim_mus_act(c), 1, 0, s0, n, 0.f);
    else return dsim_range_sum(WF(mus) + dsim_mus_act(c), 1, 0, s0, n, 0.f);
}
// ground contacts (sim.py:1137-1206) and muscle segments (sim.py:1209-1242): per-item wrenches
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_external(const Ctx& c, Exec& ex) {
    ex.mark(2);
    constexpr bool wide = DsimWideOverlap<Ctx, Exec>::value;   // contacts, segments and chunk sums ran beside the kinematics
    constexpr bool in_kin = DsimContactsInKin<Ctx, Exec::NL>::value || DsimScanFk<Ctx, Exec::NL>::value ||
                            DsimContactsAfterWalk<Ctx, Exec::NL>::value || wide;  // contacts were done by the kinematics phase
    if ((c.d.C == 0 || in_kin) && c.d.NS == 0) return;
    if constexpr (wide) return;   // ... and so did the per-body gather and the subtree sums (dsim_fwd_kinematics_wide)
    if constexpr (!wide) {
        ex.run([&](int lane) {
            if constexpr (!in_kin) {
                dsim_fwd_contacts(c, ex, lane);
                dsim_fwd_contacts_per_body(c, ex, lane);
            }
            dsim_fwd_muscle_segments(c, lane, Exec::NL);
        });
    }
    if (c.d.NS > 0) {
        // per-body gather of the muscle wrench rows in two steps (dsim_layout.hpp: mc_row): chunk sums, one lane and one LDS
        // round trip per chunk and component, then per body the sums of its chunks + its contact wrenches
        if constexpr (!wide) ex.run([&](int lane) { dsim_muscle_chunk_sums(c, lane, Exec::NL); });
        ex.run([&](int lane) {
            for (int it = lane; it < 6 * c.d.L; it += Exec::NL) {
                const int i = it / 6, k = it - 6 * i;
                float acc = dsim_body_chunk_sum(c, i, k, WF(f)[it]);
                acc = dsim_body_contact_sum(c, i, WF(cw), 6, k, acc);   // + the body's own contact wrenches
                WF(f)[it] = acc;
            }
        });
    }
}

// sum over the contacts of body i of a per-contact component (models with muscles gather per body anyway: their contact
// terms ride along in that phase, and the subtree sums that follow are sums over links only)
template <class Ctx> DSIM_FN float dsim_body_contact_sum(const Ctx& c, int i, const float* cdata, int cstride, int ck, float acc) {
    const int b0 = CI(cb_start)[i], b1 = CI(cb_start)[i + 1];
    if (c.d.flags & DSIM_F_RANGES) {
        if constexpr (DsimIsStatic<Ctx>::value)
            return dsim_range_sum_b<dsim_cap_body_contacts<decltype(c.d)>()>(cdata, cstride, ck, b0, b1 - b0, acc);
        else
            return dsim_range_sum_b<DSIM_GENERIC_BATCH>(cdata, cstride, ck, b0, b1 - b0, acc);
    }
    return dsim_gather_sum(cdata, cstride, ck, CI(cb_list), b0, b1, acc);
}
// true: contact terms are gathered per body in the muscle-gather phases; false: inside the subtree sums
template <class Ctx> DSIM_FN bool dsim_contacts_per_body(const Ctx& c) { return c.d.NS > 0; }

// sum over subtree(i) of a per-link 6-vector component + sum over the contacts of all bodies in subtree(i) of a
// per-contact component (forward: body forces + contact wrenches; adjoint: twist / pose-wrench cotangents)
template <class Ctx, class Exec>
DSIM_FN float dsim_subtree_contact_sum(const Ctx& c, Exec& ex, int lane, int i, const float* ldata, int k,
                                       const float* cdata, int cstride, int ck) {
    float acc;
    if constexpr (DsimSumMasks<Ctx, Exec::NL>::value) {
        using D = decltype(c.d);
        const DsimTopoRegs& tp = ex.topo(lane);
        acc = dsim_range_sum_m<D::L>(ldata, 6, k, i, tp.lmask, 0.f);
        if (!dsim_contacts_per_body(c)) {
            if constexpr (D::C > 0) acc = dsim_range_sum_m<D::C>(cdata, cstride, ck, tp.six_c0, tp.cmask, acc);
        }
        return acc;
    }
    if (dsim_contacts_per_body(c)) {   // contact terms are already inside the per-link data
        int n_known = -1;
        if constexpr (DsimSixRegs<Ctx, Exec::NL>::value) {
            n_known = ex.topo(lane).six_n;
            DSIM_OPAQUE(n_known);
        }
        return dsim_subtree_sum(c, ldata, 6, k, i, n_known);
    }
    if (c.d.flags & DSIM_F_RANGES) {
        DsimLinkInfo li;
        if constexpr (DsimSixRegs<Ctx, Exec::NL>::value) {
            const DsimTopoRegs& tp = ex.topo(lane);
            li.nsub = tp.six_n; li.c0 = tp.six_c0; li.nc = tp.six_nc;
            DSIM_OPAQUE(li.nsub);   // the per-entry "e < count" masks are recomputed here, not kept in (spilled) SGPR pairs
            DSIM_OPAQUE(li.nc);
        } else {
            li = dsim_link_info(c, i);
        }
        if constexpr (DsimIsStatic<Ctx>::value) {
            acc = dsim_range_sum_b<dsim_cap_links<decltype(c.d)>()>(ldata, 6, k, i, li.nsub, 0.f);
            acc = dsim_range_sum_b<dsim_cap_subtree_contacts<decltype(c.d)>()>(cdata, cstride, ck, li.c0, li.nc, acc);
        } else {
            acc = dsim_range_sum_b<DSIM_GENERIC_BATCH>(ldata, 6, k, i, li.nsub, 0.f);
            acc = dsim_range_sum_b<DSIM_GENERIC_BATCH>(cdata, cstride, ck, li.c0, li.nc, acc);
        }
    } else {
        acc = dsim_gather_sum(ldata, 6, k, CI(sub_list), CI(sub_start)[i], CI(sub_start)[i + 1], 0.f);
        acc = dsim_gather_sum(cdata, cstride, ck, CI(scb_list), CI(scb_start)[i], CI(scb_start)[i + 1], acc);
    }
    return acc;
}

// Subtree sums of a per-link 6-vector array (+ per-contact rows) for ALL links of a trunk-decomposed model, inside one phase:
//   light links   out[i] = rows [i, i + n_i) of ldata + contact rows [c0_i, c0_i + nc_i): one bounded pass, no remainder loops
//                 (n_i <= LCAP, nc_i <= CCAP), DSIM_TR_PASSES items per lane;
//   trunk links   deepest first, lanes 0..5: out[t] = ldata[t] + its own contact rows + out[children]; each step sees the
//                 previous one through the wavefront's in-order LDS queue (Exec::lds_fence: a compiler fence on the GPU).
// Replaces, for a 22-link humanoid, flat sums of up to 22 + 35 entries per item in three passes (of which the entries
// past a cap of 8 ran in run-time loops of dependent loads).  cdata may be null (links only).
// WEIGHTS: the entries of a light sum are multiplied by per-lane 1 / 0 registers (dsim_range_sum_m's trick: one fused
// multiply-add per entry instead of compare + select + add); the adjoint kernels have the registers for it (three such sums
// per substep), the forward kernel of the humanoid (one sum, 233 VGPRs already) keeps the selects.
template <class D> constexpr int dsim_trunk_pos(int link) {   // position of `link` in the trunk list, -1: a light link
    for (int u = 0; u < D::NT; ++u)
        if (D::trunk[u] == link) return u;
    return -1;
}
template <bool WEIGHTS, class Ctx, class Exec>
DSIM_FN void dsim_trunk_sum(const Ctx& c, Exec& ex, int lane, const float* ldata, const float* cdata, int cstride, int coff,
                            float* out) {
    using D = decltype(c.d);
    const DsimTopoRegs& tp = ex.topo(lane);
    constexpr int PASSES = (6 * D::NLT + Exec::NL - 1) / Exec::NL;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        int row = tp.tl_row[p], n = tp.tl_n[p], nc = tp.tl_nc[p];
        DSIM_OPAQUE(n);    // the "e < n" masks are recomputed here, not kept as loop invariants in (spilled) SGPR pairs
        DSIM_OPAQUE(nc);
        const int r0 = row < 0 ? 0 : row;
        float x[D::LCAP], y[D::CCAP > 0 ? D::CCAP : 1];
#pragma unroll
        for (int e = 0; e < D::LCAP; ++e) x[e] = ldata[r0 + 6 * e];
        const float* cp = cdata ? cdata + cstride * tp.tl_c0[p] + coff + tp.tl_k[p] : nullptr;
        if (cdata) {
#pragma unroll
            for (int e = 0; e < D::CCAP; ++e) y[e] = cp[cstride * e];
        }
        float acc = 0.f;
        if constexpr (WEIGHTS) {
#pragma unroll
            for (int e = 0; e < D::LCAP; ++e) acc = __builtin_fmaf(x[e], tp.tw_l[p][e], acc);
            if (cdata) {
#pragma unroll
                for (int e = 0; e < D::CCAP; ++e) acc = __builtin_fmaf(y[e], tp.tw_c[p][e], acc);
            }
        } else {
#pragma unroll
            for (int e = 0; e < D::LCAP; ++e) acc += (e < n) ? x[e] : 0.f;
            if (cdata) {
#pragma unroll
                for (int e = 0; e < D::CCAP; ++e) acc += (e < nc) ? y[e] : 0.f;
            }
        }
        if (row >= 0) out[row] = acc;
    }
    ex.lds_fence();
    // trunk links, deepest first, component `lane` on lanes 0..5.  The sum of a trunk child is carried in a REGISTER (ts), so
    // the chain root <- ... <- deepest trunk link is register arithmetic and everything it reads from LDS -- own rows, own
    // contacts, finished sums of the light children -- is independent of it: one LDS round trip for the whole trunk instead
    // of one per trunk link (the humanoid's four: ~250 cycles each, in four sums per env-substep).  Same terms, same order.
    if (lane < 6) {
        float ts[D::NT];
        dsim_static_for<0, D::NT>([&](auto uu) {
            constexpr int u = D::NT - 1 - decltype(uu)::value, t = D::trunk[u];
            float acc = ldata[6 * t + lane];
            if (cdata) {
#pragma unroll
                for (int e = 0; e < D::tr_ncb[u]; ++e) acc += cdata[cstride * (D::tr_cb0[u] + e) + coff + lane];
            }
            dsim_static_for<0, DSIM_TRUNK_CH>([&](auto ee) {
                constexpr int e = decltype(ee)::value;
                if constexpr (e < D::tr_nch[u]) {
                    constexpr int ch = D::tr_ch[DSIM_TRUNK_CH * u + e], pos = dsim_trunk_pos<D>(ch);
                    if constexpr (pos >= 0) acc += ts[pos];
                    else acc += out[6 * ch + lane];
                }
            });
            ts[u] = acc;
            out[6 * t + lane] = acc;
        });
    }
    ex.lds_fence();
}

// Subtree sums of a per-link 6-vector held in REGISTERS (link i on lane i, the tree on the first lanes of the wave): bottom-up by
// levels; in step s every parent adds the finished total of its child rt_d[s] lanes above (Exec::from_above: a DPP row shift --
// VALU latency, no LDS), weighted 1 / 0.  Replaces an LDS store, a phase, ~34 loads + ~34 multiply-adds on (link, component)
// lanes and the load of the result in the next phase.  Lanes that hold no link must carry zeros.  Same terms as the range sums,
// associated by tree level instead of by index: not bit-identical to them (the tests hold both to the reference).
template <class Ctx, class Exec> DSIM_FN void dsim_rowtree_sum(const Ctx&, Exec& ex, const DsimTopoRegs& tp, sv6& x) {
    using D = decltype(Ctx::d);
    // Two environments per wavefront (Exec::NL == 32): wave_shl:1 makes the LAST lane of the first environment's half read lane
    // 0 of the second environment.  Its weight there is 0 -- but 0 * Inf / NaN of a diverged neighbour is NaN -- so that lane must
    // not hold a link at all: its value is then never stored and never summed.  (dsim_pair_ok asks for L <= 16.)
    static_assert(Exec::NL >= DSIM_NL || D::L < Exec::NL, "row-tree sums with two environments per wavefront: the half's last lane must stay free");
    dsim_static_for<0, D::RT_N>([&](auto ss) {
        constexpr int s_ = decltype(ss)::value, dist = D::rt_d[s_], kind = D::rt_kind[s_];
        // The DPP steps are inline asm, which the compiler's hazard recognizer cannot see into: the s_nop in front of a step's six
        // v_fmac_f32_dpp covers "operands written by the instructions right before".  That is the case for the first step of a
        // sum and for a step that follows a FAR step (compiler-generated v_readlane + v_fma writing the same six registers); a
        // step behind another asm block reads what that block wrote five instructions earlier.
        constexpr bool GUARD = s_ == 0 || D::rt_kind[s_ > 0 ? s_ - 1 : 0] == DSIM_RT_FAR;
        // x_k += x_k[source lane] * w: a row shift, the whole-wave shift by one lane, or one far edge by v_readlane
        if constexpr (kind == DSIM_RT_ROW)
            ex.template add_from_above<dist, GUARD>(x.w.x, x.w.y, x.w.z, x.v.x, x.v.y, x.v.z, tp.rt_w[s_]);
        else if constexpr (kind == DSIM_RT_WAVE1)
            ex.template add_from_next<GUARD>(x.w.x, x.w.y, x.w.z, x.v.x, x.v.y, x.v.z, tp.rt_w[s_]);
        else
            ex.template add_from_lane<dist>(x.w.x, x.w.y, x.w.z, x.v.x, x.v.y, x.v.z, tp.rt_w[s_]);
    });
}

// value that lane `lane - SH` of the row holds (SH = 0: the lane's own): link -> dof lane of a row tree is SH = DSH, dof -> link -DSH
template <int SH, class Exec> DSIM_FN float dsim_row_shift(Exec& ex, float v) {
    if constexpr (SH > 0) return ex.template from_below<SH>(v);
    else if constexpr (SH < 0) return ex.template from_above<-SH>(v);
    else return v;
}
template <class Ctx, class Exec> struct DsimDofShift {
    static constexpr bool value = []() {
        if constexpr (DsimRowTreeFwd<Ctx, Exec>::value) return decltype(Ctx::d)::DSH_OK != 0;
        else return false;
    }();
};
// f_tot of link `lane` for row-tree models, on the link lanes: own body force + the body's contact wrenches (summed per body by the
// kinematics phase's side block), then the row-tree subtree sum.  Every lane of the wave calls it (DPP needs uniform control flow).
template <class Ctx, class Exec> DSIM_FN sv6 dsim_fwd_ftot_rowtree(const Ctx& c, Exec& ex, int lane) {
    using D = decltype(c.d);
    const int i = lane < D::L ? lane : 0;
    sv6 x = ldsv(WF(f) + 6 * i) + ldsv(WF(cwb) + 6 * i);   // (lanes without a link: link 0's values, which no step's weight selects)
    dsim_rowtree_sum(c, ex, ex.topo(lane), x);
    return x;
}

// f_tot[i] = sum over subtree(i) of (inverse-dynamics force [+ muscle wrenches, gathered per body]) + contact wrenches
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_ftot(const Ctx& c, Exec& ex) {
    ex.mark(3);
    ex.run([&](int lane) {
        if constexpr (DsimRowTreeFwd<Ctx, Exec>::value) {
            const sv6 t = dsim_fwd_ftot_rowtree(c, ex, lane);
            if (lane < c.d.L) stsv(WF(ftot) + 6 * lane, t);
            return;
        }
        if constexpr (DsimTrunk<Ctx, Exec>::value) {
            dsim_trunk_sum<true>(c, ex, lane, WF(f), WF(cw), 6, 0, WF(ftot));
            return;
        }
        for (int it = lane; it < 6 * c.d.L; it += Exec::NL) {
            const int i = it / 6, k = it - 6 * i;
            WF(ftot)[it] = dsim_subtree_contact_sum(c, ex, lane, i, WF(f), k, WF(cw), 6, k);
        }
    });
}
// joint-space forces (sim.py:1421-1502, 1792-1842)
// joint-space forces of the dofs `lane`, `lane + NL`, ... from f_tot (one lane per dof)
template <class Ctx, class Exec> DSIM_FN void dsim_tau_lane(const Ctx& c, Exec& ex, int lane) {
    for (int d = lane; d < c.d.nd; d += Exec::NL) {
        int i, type, cs, ds;
        if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
            const DsimTopoRegs& tp = ex.topo(lane);
            i = tp.dof_link; type = tp.dof_type; cs = tp.dof_cs; ds = tp.dof_ds;
            DSIM_OPAQUE(type);
        } else {
            i = CI(dof_link)[d]; type = CI(jtype)[i]; cs = CI(qstart)[i]; ds = CI(qdstart)[i];
        }
        float t = 0.0f - sdot(ldsv(WF(S) + 6 * d), ldsv(WF(ftot) + 6 * i));
        if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
            const float q = WF(q)[cs], qd = WF(qd)[d];
            const float lower = CF(lower)[cs], upper = CF(upper)[cs], lke = CF(lke)[i];
            float limit_f = 0.0f;
            if (q < lower) limit_f = lke * (lower - q);
            if (q > upper) limit_f = lke * (upper - q);
            t = t - CF(tke)[i] * (q - CF(target)[cs]) - CF(tkd)[i] * qd + WF(act)[d] + limit_f - CF(lkd)[i] * qd;
        } else if (type == DSIM_JOINT_BALL) {
            const int k = d - ds;
            t = t - WF(qd)[d] * CF(tkd)[i] - WF(q)[cs + k] * CF(tke)[i];
        }
        WF(tau)[d] = t;
    }
}
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_tau(const Ctx& c, Exec& ex) {
    if constexpr (!DsimWideOverlap<Ctx, Exec>::value) dsim_fwd_ftot(c, ex);   // (several wavefronts: summed behind the kinematics)
    ex.run([&](int lane) { dsim_tau_lane(c, ex, lane); });
}

// composite inertias Ic[i] = sum over subtree(i), F_b = Ic[link(b)] S_b
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_composite(const Ctx& c, Exec& ex) {
    ex.mark(4);
    const int nd = c.d.nd;
    ex.run([&](int lane) {
        for (int it = lane; it < 10 * c.d.L; it += Exec::NL) {
            const int i = it / 10, k = it - 10 * i;
            WF(ic10)[it] = dsim_subtree_sum(c, WF(i10), 10, k, i);
        }
    });
    ex.run([&](int lane) {
        for (int b = lane; b < nd; b += Exec::NL) {
            const inertia10 I = ld_i10(WF(ic10) + 10 * CI(dof_link)[b]);
            stsv(WF(F) + 6 * b, inertia_mul(I, ldsv(WF(S) + 6 * b)));
        }
    });
}

// the register / cross-lane form of the Gauss-Jordan inverse needs a compile-time matrix size that fits one wavefront's lanes
template <class Ctx, class Exec> struct DsimWaveGj {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) return decltype(Ctx::d)::nd <= 32;
        else return false;
    }();
};
// the mass matrix filled from its upper triangle (specialised kernels; -DDSIM_NO_SYM_FILL builds round 5's full fill for A/B runs)
// One-wave mappings only: measured (tools/ab_min.py, profiles/r06_experiments.txt item 7) Ant forward -1.5 %, Humanoid -1.6 %; the
// four-wave SNUHumanoid kernel +5 % (576 entries are 3 passes of its 256 lanes already; the folded index arithmetic and the
// second store cost it more than the third pass did).
template <class Ctx, class Exec> struct DsimSymFill {
    static constexpr bool value = []() {
#ifdef DSIM_NO_SYM_FILL
        return false;
#else
        return DsimIsStatic<Ctx>::value && Exec::NL <= DSIM_NL;
#endif
    }();
};
// H = J^T M J in composite-rigid-body form + armature, inverted in place (Gauss-Jordan, SPD, no pivoting)
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_mass(const Ctx& c, Exec& ex) {
    const int nd = c.d.nd;
    dsim_fwd_composite(c, ex);
    if constexpr (DsimSymFill<Ctx, Exec>::value) {
        // H is symmetric: the entries on and above the diagonal are evaluated (nd (nd + 1) / 2 items instead of nd^2: Ant 2 passes
        // of a wavefront instead of 4, Humanoid 7 instead of 12) and stored twice.  The upper triangle is dealt to the lanes as a
        // folded rectangle -- row r (nd - r entries) and row nd - 1 - r (r + 1 entries) share one row of nd + 1 items -- so that
        // an item's (a, b) is a division by a compile-time constant and two selects.  Dofs are numbered parents first, so a <= b
        // never has link(a) strictly below link(b): rel is 0 or 1 there.  (The entries below the diagonal of one link's own dofs
        // used to be S_b . (Ic S_a) instead of S_a . (Ic S_b): equal numbers, rounded differently; now H is symmetric to the bit.)
        constexpr int ND = decltype(c.d)::nd, NDE = ND + (ND & 1), W = NDE + 1, NT = (NDE / 2) * W;
        ex.run([&](int lane) {
#pragma unroll
            for (int p = 0; p < (NT + Exec::NL - 1) / Exec::NL; ++p) {
                const int it = lane + Exec::NL * p;
                const int r = it / W, cc = it - W * r;
                const bool lo = cc < NDE - r;
                const int a = lo ? r : NDE - 1 - r, b = lo ? r + cc : (NDE - 1 - r) + (cc - (NDE - r));
                const bool on = it < NT && b < ND;   // (odd nd: the folded rectangle of nd + 1 has a spare row and column)
                const int ai = on ? a : 0, bi = on ? b : 0;
                const int rl = CI(rel)[ai * ND + bi];
                const sv6 Sa = ldsv(WF(S) + 6 * ai), Fb = ldsv(WF(F) + 6 * bi);
                const float arm = CF(arm)[ai];
                float hv = rl != 0 ? sdot(Sa, Fb) : 0.f;
                if (a == b) hv += arm;
                if (on) {
                    WF(hinv)[a * ND + b] = hv;
                    WF(hinv)[b * ND + a] = hv;
                }
            }
        });
    } else
    ex.run([&](int lane) {
        for (int it = lane; it < nd * nd; it += Exec::NL) {
            const int a = it / nd, b = it - nd * a;
            const int r = CI(rel)[it];
            float hv = 0.f;
            if (r == 1) hv = sdot(ldsv(WF(S) + 6 * a), ldsv(WF(F) + 6 * b));
            else if (r == 2) hv = sdot(ldsv(WF(S) + 6 * b), ldsv(WF(F) + 6 * a));
            if (a == b) hv += CF(arm)[a];
            WF(hinv)[it] = hv;
        }
    });
    // In-place Gauss-Jordan inverse, no pivoting (H is SPD).  Per pivot k: p_j = (j == k ? 1 : H[k][j]) * (1 / H[k][k]);
    // row k <- p; every other row i: H[i][j] <- (j == k ? 0 : H[i][j]) - H[i][k] p_j.
    // Specialised kernels: ONE phase -- lane i keeps row i in registers and the pivot row is broadcast across the
    // wavefront with v_readlane (Exec::wave_gj), instead of 2 nd phases of LDS round trips.  Same formulas, same order.
    if constexpr (DsimWaveGj<Ctx, Exec>::value) {
        ex.template wave_gj<decltype(c.d)::nd>(WF(hinv));
    } else if (Exec::WAVE_GJ_PAD && nd <= 32) {
        // generic kernels: the same register inversion on the matrix padded to 16 or 32 (dsim_hip.hip: dsim_wave_gj_pad)
        if (nd <= 16) ex.template wave_gj_pad<16>(WF(hinv), nd);
        else ex.template wave_gj_pad<32>(WF(hinv), nd);
    } else {
        for (int k = 0; k < nd; ++k) {
            ex.run([&](int lane) {
                for (int j = lane; j < nd; j += Exec::NL) {
                    const float rp = 1.0f / WF(hinv)[k * nd + k];
                    WF(prow)[j] = (j == k ? 1.0f : WF(hinv)[k * nd + j]) * rp;
                    WF(pcol)[j] = WF(hinv)[j * nd + k];
                }
            });
            ex.run([&](int lane) {
                for (int it = lane; it < nd * nd; it += Exec::NL) {
                    const int i = it / nd, j = it - nd * i;
                    if (i == k) WF(hinv)[it] = WF(prow)[j];
                    else WF(hinv)[it] = (j == k ? 0.0f : WF(hinv)[it]) - WF(pcol)[i] * WF(prow)[j];
                }
            });
        }
    }
}

template <class Ctx, class Exec> DSIM_FN void dsim_fwd_solve(const Ctx& c, Exec& ex) {
    ex.mark(5);
    const int nd = c.d.nd;
    ex.run([&](int lane) {
        for (int i = lane; i < nd; i += Exec::NL) {
            WF(qdd)[i] = dsim_dot_n(WF(hinv) + i * nd, WF(tau), nd);
        }
    });
}

// joint-type mask of the whole model: a compile-time constant in the specialised kernels
template <class Ctx> DSIM_FN constexpr int dsim_tmask_static() {
    if constexpr (DsimIsStatic<Ctx>::value) return decltype(Ctx::d)::tmask;
    else return 0x1f;
}

// semi-implicit Euler (sim.py:1505-1636); in place on q, qd.  Loads first, then arithmetic, then stores.
template <class Ctx, class Exec> DSIM_FN void dsim_integrate_lane(const Ctx& c, Exec& ex, int lane) {
    constexpr int MASK = dsim_tmask_static<Ctx>();
    constexpr int NQ = dsim_mask_nq(MASK), NDF = dsim_mask_nd(MASK);
    const float h = c.h;
    for (int i = lane; i < c.d.L; i += Exec::NL) {
        int type, cs, ds;
        if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
            const DsimTopoRegs& tp = ex.topo(lane);
            type = tp.own_type; cs = tp.own_cs; ds = tp.own_ds;
            DSIM_OPAQUE(type);
        } else {
            type = CI(jtype)[i]; cs = CI(qstart)[i]; ds = CI(qdstart)[i];
        }
        float *q = WF(q), *qd = WF(qd);
        const float* qdd = WF(qdd);
        float qv[NQ > 0 ? NQ : 1], qdv[NDF > 0 ? NDF : 1], av[NDF > 0 ? NDF : 1];
#pragma unroll
        for (int k = 0; k < NQ; ++k) qv[k] = q[cs + k];
#pragma unroll
        for (int k = 0; k < NDF; ++k) {
            qdv[k] = qd[ds + k];
            av[k] = qdd[ds + k];
        }
        if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
            const float qdn = qdv[0] + av[0] * h;
            qd[ds] = qdn;
            q[cs] = qv[0] + qdn * h;
        }
        if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
            if (type == DSIM_JOINT_BALL || type == DSIM_JOINT_FREE) {
                const bool fr = type == DSIM_JOINT_FREE;
                const v3 w = mk3(qdv[0], qdv[1], qdv[2]) + mk3(av[0], av[1], av[2]) * h;
                q4 r;
                v3 pn = zero3(), vn = zero3();
                if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                    if (fr) {
                        vn = mk3(qdv[3], qdv[4], qdv[5]) + mk3(av[3], av[4], av[5]) * h;
                        const v3 p = mk3(qv[0], qv[1], qv[2]);
                        pn = p + (vn + cross(w, p)) * h;
                        r = mkq(qv[3], qv[4], qv[5], qv[6]);
                    } else {
                        r = mkq(qv[0], qv[1], qv[2], qv[3]);
                    }
                } else {
                    r = mkq(qv[0], qv[1], qv[2], qv[3]);
                }
                const q4 dr = qmul_v(w, r) * 0.5f;
                const q4 rt = r + dr * h;
                const float il = dsim_inv_len(qdot(rt, rt));
                q4 rn = mkq(0.f, 0.f, 0.f, 1.f);
                if (il > 0.0f) rn = rt * il;
                if (fr) {
                    st3(q + cs, pn);
                    st3(qd + ds + 3, vn);
                }
                stq(q + cs + (fr ? 3 : 0), rn);
                st3(qd + ds, w);
            }
        }
    }
}
template <class Ctx, class Exec> DSIM_FN void dsim_fwd_integrate(const Ctx& c, Exec& ex) {
    ex.mark(6);
    ex.run([&](int lane) { dsim_integrate_lane(c, ex, lane); });
}

// ---- forward dynamics of a substep in ONE phase (small models, one wavefront per environment) -----------------------------
// f_tot (subtree sums) -> tau -> qdd = H^-1 tau -> checkpoint copy -> integrate used to be four phases whose only
// connection is a handful of values that change lanes: the 6 components of f_tot[link(d)] go from the (link, component)
// lanes to dof lane d, tau_j goes from dof lane j to every dof lane, qdd of a link's dofs goes to the link's lane.  With
// the wavefront's cross-lane primitives (Exec::shfl = ds_bpermute, Exec::bcast = v_readlane) these exchanges stay in
// registers, and three LDS store -> phase boundary -> load round trips disappear (forward substep of Ant: 5 phases -> 2).
// Same arithmetic in the same order as dsim_fwd_tau / dsim_fwd_solve / dsim_fwd_integrate.
template <class Ctx, class Exec> struct DsimWaveDyn {
    static constexpr bool value = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value && Exec::WAVE_OPS) {
            using D = decltype(Ctx::d);
            return D::L <= Exec::NL && D::nd <= 32 && D::NS == 0 && (D::flags & DSIM_F_RANGES) != 0;
        } else {
            return false;
        }
    }();
    // ... with the subtree sums inside the phase too, on (link, component) lanes (6 L lanes); models with more links sum in a
    // phase of their own (dsim_fwd_ftot: trunk decomposition) and the dof lanes read f_tot of their links from LDS
    static constexpr bool sums_inside = []() {
        if constexpr (std::is_empty<decltype(Ctx::d)>::value) return 6 * decltype(Ctx::d)::L <= Exec::NL;
        else return false;
    }();
};
// The integrator's 1 / |r + dr h| of the free root's quaternion update rides in the saved block (DsimOff::qil, a padding word
// behind qdd): integrate^T multiplies by it instead of redoing the square root and the division on its one busy lane (~35
// dependent instructions at the head of every adjoint substep).  Full checkpoint mode, models whose only quaternion joint is
// the free root (Ant, Humanoid); everything else recomputes as before.
template <class Ctx, class Exec> struct DsimSavedIl {
    static constexpr bool value = []() {
        if constexpr (DsimWaveDyn<Ctx, Exec>::value && !Ctx::LEAN) {
            using D = decltype(Ctx::d);
            bool ok = decltype(Ctx::o)::qil >= 0 && (D::tmask & DSIM_TM(DSIM_JOINT_BALL)) == 0 && (D::pmask[0] & DSIM_TM(DSIM_JOINT_FREE)) != 0 &&
                      D::D <= DSIM_PMASK_N;
            for (int p = 1; p < DSIM_PMASK_N; ++p) ok = ok && (D::pmask[p] & DSIM_TM(DSIM_JOINT_FREE)) == 0;
            return ok;
        } else {
            return false;
        }
    }();
};
template <class Ctx, class Exec>
DSIM_FN void dsim_fwd_dynamics_wave(const Ctx& c, Exec& ex, float* g_row, float* g_hinv, bool update_mass) {
    ex.mark(3);
    using D = decltype(c.d);
    constexpr int nd = D::nd, L = D::L;
    // The detached side block below (checkpoint copy on the helper wavefront) is still reading X_sc .. qdd and H^-1 when the
    // main wave returns: the NEXT substep's kinematics must therefore start behind a workgroup barrier, which only the
    // fork_join / fork_join_mid forms of dsim_fwd_kinematics have.  A specialised model with a helper wave whose tree is deeper
    // than DSIM_CHAIN_MAX / 2^DSIM_SCAN_ROUNDS_MAX would fall through to the plain ex.run() walk (no barrier) and overwrite
    // the rows under the copy: refuse to compile that combination instead of corrupting checkpoints.
    static_assert(!Exec::HAS_HELPER || DsimScanFk<Ctx, Exec::NL>::value || DsimContactsAfterWalk<Ctx, Exec::NL>::value ||
                      DsimContactsInKin<Ctx, Exec::NL>::value,
                  "helper-wave kernels need a kinematics phase that starts with a workgroup barrier (tree too deep for the chain "
                  "registers and the scan): build this model without a helper wavefront (dsim_has_helper)");
    ex.fork_mid_detached([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        constexpr int NQ = dsim_mask_nq(MASK), NDF = dsim_mask_nd(MASK);
        const float h = c.h;
        const DsimTopoRegs& tp = ex.topo(lane);
        const bool is_dof = lane < nd, is_link = lane < L;
        // ---- loads of all three roles, issued together
        // dof role
        int dtype = tp.dof_type;
        DSIM_OPAQUE(dtype);
        const int di = tp.dof_link, dcs = tp.dof_cs, dds = tp.dof_ds, d = is_dof ? lane : 0;
        const bool hinge = dtype == DSIM_JOINT_PRISMATIC || dtype == DSIM_JOINT_REVOLUTE;
        const int qi = hinge ? dcs : (dtype == DSIM_JOINT_BALL ? dcs + (d - dds) : 0);
        const sv6 Sd = ldsv(WF(S) + 6 * d);
        const float q_d = WF(q)[qi], qd_d = WF(qd)[d], act_d = WF(act)[d];
        const float lower = CF(lower)[qi], upper = CF(upper)[qi], target = CF(target)[qi];
        const float lke = CF(lke)[di], tke = CF(tke)[di], tkd = CF(tkd)[di], lkd = CF(lkd)[di];
        float hrow[nd];
#pragma unroll
        for (int j = 0; j < nd; ++j) hrow[j] = WF(hinv)[d * nd + j];
        // link role
        int ltype = tp.own_type;
        DSIM_OPAQUE(ltype);
        const int lcs = tp.own_cs, lds_ = tp.own_ds;
        float qv[NQ > 0 ? NQ : 1], qdv[NDF > 0 ? NDF : 1];
#pragma unroll
        for (int k = 0; k < NQ; ++k) qv[k] = WF(q)[lcs + k];
#pragma unroll
        for (int k = 0; k < NDF; ++k) qdv[k] = WF(qd)[lds_ + k];
        sv6 F;
        if constexpr (DsimRowTreeFwd<Ctx, Exec>::value && DsimWaveDyn<Ctx, Exec>::sums_inside) {
            // ---- link role: f_tot = row-tree subtree sums on registers; f_tot[link(d)] -> dof lane d
            const sv6 ftl = dsim_fwd_ftot_rowtree(c, ex, lane);
            if (is_link) stsv(WF(ftot) + 6 * lane, ftl);   // the adjoint reads it from the checkpoint
            ex.stamp();
            if constexpr (DsimDofShift<Ctx, Exec>::value) {
                // one row shift instead of six ds_bpermute round trips: dof lane d holds link d - DSH's total; the root's dofs
                // take lane 0's by v_readlane
                F.w.x = dsim_row_shift<D::DSH>(ex, ftl.w.x); F.w.y = dsim_row_shift<D::DSH>(ex, ftl.w.y); F.w.z = dsim_row_shift<D::DSH>(ex, ftl.w.z);
                F.v.x = dsim_row_shift<D::DSH>(ex, ftl.v.x); F.v.y = dsim_row_shift<D::DSH>(ex, ftl.v.y); F.v.z = dsim_row_shift<D::DSH>(ex, ftl.v.z);
                if constexpr (D::ND_ROOT > 0) {
                    const sv6 Fr = mksv(mk3(ex.bcast(ftl.w.x, 0), ex.bcast(ftl.w.y, 0), ex.bcast(ftl.w.z, 0)),
                                        mk3(ex.bcast(ftl.v.x, 0), ex.bcast(ftl.v.y, 0), ex.bcast(ftl.v.z, 0)));
                    if (lane < D::ND_ROOT) F = Fr;
                }
            } else {
                F.w.x = ex.shfl(ftl.w.x, di); F.w.y = ex.shfl(ftl.w.y, di); F.w.z = ex.shfl(ftl.w.z, di);
                F.v.x = ex.shfl(ftl.v.x, di); F.v.y = ex.shfl(ftl.v.y, di); F.v.z = ex.shfl(ftl.v.z, di);
            }
        } else if constexpr (DsimWaveDyn<Ctx, Exec>::sums_inside) {
            // ---- (link, component) role: f_tot = subtree sums of the body forces and contact wrenches
            float ft = 0.f;
            if (lane < 6 * L) {
                ft = dsim_subtree_contact_sum(c, ex, lane, lane / 6, WF(f), lane - 6 * (lane / 6), WF(cw), 6, lane - 6 * (lane / 6));
                WF(ftot)[lane] = ft;   // the adjoint reads it from the checkpoint
            }
            ex.stamp();
            // ---- f_tot[link(d)] -> dof lane d
            F.w.x = ex.shfl(ft, 6 * di + 0); F.w.y = ex.shfl(ft, 6 * di + 1); F.w.z = ex.shfl(ft, 6 * di + 2);
            F.v.x = ex.shfl(ft, 6 * di + 3); F.v.y = ex.shfl(ft, 6 * di + 4); F.v.z = ex.shfl(ft, 6 * di + 5);
        } else {
            F = ldsv(WF(ftot) + 6 * di);   // summed by the phase before (dsim_fwd_ftot)
        }
        // ---- dof role: tau (sim.py:1421-1502)
        float t = 0.0f - sdot(Sd, F);
        if (hinge) {
            float limit_f = 0.0f;
            if (q_d < lower) limit_f = lke * (lower - q_d);
            if (q_d > upper) limit_f = lke * (upper - q_d);
            t = t - tke * (q_d - target) - tkd * qd_d + act_d + limit_f - lkd * qd_d;
        } else if (dtype == DSIM_JOINT_BALL) {
            t = t - qd_d * tkd - q_d * tke;
        }
        // ---- qdd = H^-1 tau: tau_j travels by v_readlane
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < nd; ++j) acc += hrow[j] * ex.bcast(t, j);
        if (is_dof) {
            WF(tau)[lane] = t;
            WF(qdd)[lane] = acc;
        }
        ex.stamp();
        // ---- qdd of a link's dofs -> the link's lane
        float av[NDF > 0 ? NDF : 1];
        if constexpr (DsimDofShift<Ctx, Exec>::value) {
            // link lane l <- dof lane l + DSH (its one dof); the root <- its ND_ROOT dofs on lanes 0 .. ND_ROOT - 1 (v_readlane)
#pragma unroll
            for (int k = 0; k < NDF; ++k) av[k] = (k < D::ND_ROOT) ? ex.bcast(acc, k < D::ND_ROOT ? k : 0) : 0.f;
            const float own = dsim_row_shift<-D::DSH>(ex, acc);
            if (lane > 0 || D::ND_ROOT == 0) av[0] = own;
        } else {
#pragma unroll
            for (int k = 0; k < NDF; ++k) av[k] = ex.shfl(acc, lds_ + k);
        }
        // ---- link role: semi-implicit Euler (sim.py:1505-1636).  The arithmetic comes first -- the inverse norm of the root's
        // quaternion update belongs to the saved block (DsimSavedIl) -- then the hand-over of the checkpoint copy, then the stores
        float qdn = 0.f, qn = 0.f;
        v3 w = zero3(), pn = zero3(), vn = zero3();
        q4 rn = mkq(0.f, 0.f, 0.f, 1.f);
        const bool hinge_l = ltype == DSIM_JOINT_PRISMATIC || ltype == DSIM_JOINT_REVOLUTE;
        const bool quat_l = ltype == DSIM_JOINT_BALL || ltype == DSIM_JOINT_FREE, fr = ltype == DSIM_JOINT_FREE;
        if (hinge_l) {
            qdn = qdv[0] + av[0] * h;
            qn = qv[0] + qdn * h;
        }
        if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
            if (quat_l) {
                w = mk3(qdv[0], qdv[1], qdv[2]) + mk3(av[0], av[1], av[2]) * h;
                q4 r;
                if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                    if (fr) {
                        vn = mk3(qdv[3], qdv[4], qdv[5]) + mk3(av[3], av[4], av[5]) * h;
                        const v3 p = mk3(qv[0], qv[1], qv[2]);
                        pn = p + (vn + cross(w, p)) * h;
                        r = mkq(qv[3], qv[4], qv[5], qv[6]);
                    } else {
                        r = mkq(qv[0], qv[1], qv[2], qv[3]);
                    }
                } else {
                    r = mkq(qv[0], qv[1], qv[2], qv[3]);
                }
                const q4 dr = qmul_v(w, r) * 0.5f;
                const q4 rt = r + dr * h;
                const float il = dsim_inv_len(qdot(rt, rt));
                if (il > 0.0f) rn = rt * il;
                if constexpr (DsimSavedIl<Ctx, Exec>::value) {
                    if (is_link && fr) WF(qil)[0] = il;
                }
            }
        }
        // ---- f_tot and qdd (and the word above) of all lanes are in LDS: the rest of the checkpoint row may be copied (side block
        // below; its head -- q, qd, which the stores below overwrite -- went out during the kinematics)
        ex.mid();
        ex.stamp();
        if (is_link) {
            float *q = WF(q), *qd = WF(qd);
            if (hinge_l) {
                qd[lds_] = qdn;
                q[lcs] = qn;
            }
            if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
                if (quat_l) {
                    if (fr) {
                        st3(q + lcs, pn);
                        st3(qd + lds_ + 3, vn);
                    }
                    stq(q + lcs + (fr ? 3 : 0), rn);
                    st3(qd + lds_, w);
                }
            }
        }
    }, [&](int lane) {
        // side block (helper wavefront where there is one): the rest of the checkpoint row and, on a refresh, the inverse.  Nothing
        // it reads is written again before the next substep's kinematics, which start behind a barrier both waves take.
        if (g_row) {
            dsim_ckpt_store_row<Ctx, Exec::NL, 2>(c, lane, g_row);
            if (update_mass && g_hinv) {
                constexpr int IT = (nd * nd + Exec::NL - 1) / Exec::NL;
                float x[IT];
#pragma unroll
                for (int r = 0; r < IT; ++r) x[r] = WF(hinv)[(lane + Exec::NL * r) < nd * nd ? lane + Exec::NL * r : 0];
#pragma unroll
                for (int r = 0; r < IT; ++r)
                    if (lane + Exec::NL * r < nd * nd) g_hinv[lane + Exec::NL * r] = x[r];
            }
        }
    });
}

// ---- the same for a model with several wavefronts per environment (DsimWideOverlap; round 6) -------------------------------
// tau -> H^-1 tau -> integrate were three phases of all four wavefronts (three workgroup barriers, two LDS store -> load round
// trips; 24 dofs and 11 links busy, 192 lanes waiting: 2.9 k of the 11 k cycles of a SNUHumanoid substep).  Now ONE block of the
// FIRST wavefront, the values changing lanes in registers (tau_j by v_readlane, qdd of a link's dofs by ds_bpermute), while the
// others wait at ONE hand-over (mid: f_tot, tau and qdd are in LDS) and then copy the rest of the checkpoint row -- X_sc .. qdd,
// which the integrator's stores do not touch -- beside the integrator.  Same arithmetic in the same order as dsim_tau_lane /
// dsim_fwd_solve / dsim_integrate_lane.  (-DDSIM_NO_WIDE_DYN builds the three-phase form for A/B runs.)
template <class Ctx, class Exec> struct DsimWideDyn {
    static constexpr bool value = []() {
#ifdef DSIM_NO_WIDE_DYN
        return false;
#else
        if constexpr (DsimWideOverlap<Ctx, Exec>::value) return decltype(Ctx::d)::nd <= 32 && decltype(Ctx::d)::L <= DSIM_NL;
        else return false;
#endif
    }();
};
template <class Ctx, class Exec>
DSIM_FN void dsim_fwd_dynamics_wide(const Ctx& c, Exec& ex, float* g_row, float* g_hinv, bool update_mass) {
    ex.mark(3);
    using D = decltype(c.d);
    constexpr int nd = D::nd, L = D::L, NLR = Exec::NL - DSIM_NL;
    ex.fork_wave0([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        constexpr int NQ = dsim_mask_nq(MASK), NDF = dsim_mask_nd(MASK);
        const float h = c.h;
        const DsimTopoRegs& tp = ex.topo(lane);
        const bool is_dof = lane < nd, is_link = lane < L;
        // ---- loads of both roles, issued together
        int dtype = tp.dof_type;
        DSIM_OPAQUE(dtype);
        const int di = tp.dof_link, dcs = tp.dof_cs, dds = tp.dof_ds, d = is_dof ? lane : 0;
        const bool hinge = dtype == DSIM_JOINT_PRISMATIC || dtype == DSIM_JOINT_REVOLUTE;
        const int qi = hinge ? dcs : (dtype == DSIM_JOINT_BALL ? dcs + (d - dds) : 0);
        const sv6 Sd = ldsv(WF(S) + 6 * d);
        const sv6 F = ldsv(WF(ftot) + 6 * di);   // summed behind the kinematics by this wavefront
        const float q_d = WF(q)[qi], qd_d = WF(qd)[d], act_d = WF(act)[d];
        const float lower = CF(lower)[qi], upper = CF(upper)[qi], target = CF(target)[qi];
        const float lke = CF(lke)[di], tke = CF(tke)[di], tkd = CF(tkd)[di], lkd = CF(lkd)[di];
        float hrow[nd];
#pragma unroll
        for (int j = 0; j < nd; ++j) hrow[j] = WF(hinv)[d * nd + j];
        int ltype = tp.own_type;
        DSIM_OPAQUE(ltype);
        const int lcs = tp.own_cs, lds_ = tp.own_ds;
        float qv[NQ > 0 ? NQ : 1], qdv[NDF > 0 ? NDF : 1];
#pragma unroll
        for (int k = 0; k < NQ; ++k) qv[k] = WF(q)[lcs + k];
#pragma unroll
        for (int k = 0; k < NDF; ++k) qdv[k] = WF(qd)[lds_ + k];
        // ---- dof role: tau (sim.py:1421-1502)
        float t = 0.0f - sdot(Sd, F);
        if (hinge) {
            float limit_f = 0.0f;
            if (q_d < lower) limit_f = lke * (lower - q_d);
            if (q_d > upper) limit_f = lke * (upper - q_d);
            t = t - tke * (q_d - target) - tkd * qd_d + act_d + limit_f - lkd * qd_d;
        } else if (dtype == DSIM_JOINT_BALL) {
            t = t - qd_d * tkd - q_d * tke;
        }
        // ---- qdd = H^-1 tau: tau_j travels by v_readlane (the order of dsim_dot_n: j ascending, acc += a_j b_j)
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < nd; ++j) acc += hrow[j] * ex.bcast(t, j);
        if (is_dof) {
            WF(tau)[lane] = t;
            WF(qdd)[lane] = acc;
        }
        // ---- qdd of a link's dofs -> the link's lane
        float av[NDF > 0 ? NDF : 1];
#pragma unroll
        for (int k = 0; k < NDF; ++k) av[k] = ex.shfl(acc, lds_ + k);
        // ---- link role: semi-implicit Euler (sim.py:1505-1636); arithmetic, hand-over of the checkpoint copy, stores
        float qdn = 0.f, qn = 0.f;
        v3 w = zero3(), pn = zero3(), vn = zero3();
        q4 rn = mkq(0.f, 0.f, 0.f, 1.f);
        const bool hinge_l = ltype == DSIM_JOINT_PRISMATIC || ltype == DSIM_JOINT_REVOLUTE;
        const bool quat_l = ltype == DSIM_JOINT_BALL || ltype == DSIM_JOINT_FREE, fr = ltype == DSIM_JOINT_FREE;
        if (hinge_l) {
            qdn = qdv[0] + av[0] * h;
            qn = qv[0] + qdn * h;
        }
        if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
            if (quat_l) {
                w = mk3(qdv[0], qdv[1], qdv[2]) + mk3(av[0], av[1], av[2]) * h;
                q4 r;
                if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                    if (fr) {
                        vn = mk3(qdv[3], qdv[4], qdv[5]) + mk3(av[3], av[4], av[5]) * h;
                        const v3 p = mk3(qv[0], qv[1], qv[2]);
                        pn = p + (vn + cross(w, p)) * h;
                        r = mkq(qv[3], qv[4], qv[5], qv[6]);
                    } else {
                        r = mkq(qv[0], qv[1], qv[2], qv[3]);
                    }
                } else {
                    r = mkq(qv[0], qv[1], qv[2], qv[3]);
                }
                const q4 dr = qmul_v(w, r) * 0.5f;
                const q4 rt = r + dr * h;
                const float il = dsim_inv_len(qdot(rt, rt));
                if (il > 0.0f) rn = rt * il;
            }
        }
        ex.mid();   // tau, qdd of all dofs are in LDS (f_tot since the kinematics): the others copy the rest of the row
        if (is_link) {
            float *q = WF(q), *qd = WF(qd);
            if (hinge_l) {
                qd[lds_] = qdn;
                q[lcs] = qn;
            }
            if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
                if (quat_l) {
                    if (fr) {
                        st3(q + lcs, pn);
                        st3(qd + lds_ + 3, vn);
                    }
                    stq(q + lcs + (fr ? 3 : 0), rn);
                    st3(qd + lds_, w);
                }
            }
        }
    }, [&](int lane) {
        ex.mid();
        if (g_row) {
            dsim_ckpt_store_row<Ctx, NLR, 2>(c, lane, g_row);
            if (update_mass && g_hinv)
                for (int k = lane; k < nd * nd; k += NLR) g_hinv[k] = WF(hinv)[k];
        }
    });
}

// ---- checkpoint = what the adjoint launch needs from the forward launch, kept in HBM instead of recomputed ----
// per environment: [substeps][save_words] saved blocks (q, qd, X_sj, X_sc, COM, S, v_j, v, a, inertias, f_tot, qdd of the
// substep) followed by [groups][hinv_words] inverses of the mass matrix (one per refresh).  288 GB of HBM are otherwise
// idle on this path (Ant: 37 KB per env-step), and reading the block back costs ~1.5k cycles against ~10k to recompute it.
DSIM_FN int dsim_hinv_words_d(int nd) { return (nd * nd + 3) & ~3; }
template <class Ctx> DSIM_FN float* dsim_ckpt_hinv(const Ctx& c, float* g_ckpt, int substeps, int group) {
    return g_ckpt + (size_t)substeps * dsim_row(c) + (size_t)group * dsim_hinv_words_d(c.d.nd);
}
// tail of an environment's checkpoint: [q, qd at the end of the step (before any episode reset), episode flags]
template <class Ctx> DSIM_FN float* dsim_ckpt_tail(const Ctx& c, float* g_ckpt, int substeps, int mm_freq) {
    const int groups = (substeps + mm_freq - 1) / mm_freq;
    return g_ckpt + (size_t)substeps * dsim_row(c) + (size_t)groups * dsim_hinv_words_d(c.d.nd);
}

// Precondition of the whole path (include/dsim.h, dsim_step_forward): every quaternion block of the state handed in is a UNIT
// quaternion.  The kinematics, the 10-parameter inertia and the wrench form of the adjoint are rotations only there (the
// reference evaluates its formulas literally off the manifold, quat.h:113-116, and gives something else: measured 5e-3 in qd
// at |q| = 1.001).  Checked once per launch on the state as loaded: | |q|^2 - 1 | > 2e-4 (1e-4 in the norm) stores
// (1, environment) into the model's status words, which live in host memory mapped into the device; the NEXT call on the
// model returns DSIM_ERR_INVALID.  The integrator renormalises every substep (sim.py:1552, 1616), so states produced by the
// path itself always pass.  One phase of a few instructions per launch; models without quaternion joints compile nothing.
#define DSIM_UNIT_QUAT_TOL 2e-4f
template <class Ctx, class Exec> DSIM_FN void dsim_check_unit_quats(const Ctx& c, Exec& ex, int* g_status, int env) {
    constexpr int MASK = dsim_tmask_static<Ctx>();
    if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
        if (!g_status) return;
        ex.fire([&](int lane) {
            for (int i = lane; i < c.d.L; i += Exec::NL) {
                const int type = CI(jtype)[i], cs = CI(qstart)[i];
                if (type != DSIM_JOINT_BALL && type != DSIM_JOINT_FREE) continue;
                const q4 r = ldq(WF(q) + cs + (type == DSIM_JOINT_FREE ? 3 : 0));
                if (!(fabsf(qdot(r, r) - 1.0f) <= DSIM_UNIT_QUAT_TOL)) {   // (also catches NaN)
                    g_status[1] = env;
                    ex.system_fence();   // the environment index is in host memory before the flag that makes the host read it
                    g_status[0] = 1;
                }
            }
        });
    }
}

// one substep on the LDS-resident state; g_row / g_hinv: where to stream the saved block / the fresh inverse (or null)
template <class Ctx, class Exec>
DSIM_FN void dsim_fwd_substep(const Ctx& c, Exec& ex, bool update_mass, float* g_row = nullptr, float* g_hinv = nullptr) {
    constexpr bool wide = DsimWideOverlap<Ctx, Exec>::value;
    dsim_fwd_kinematics(c, ex, (DsimWaveDyn<Ctx, Exec>::value || wide) ? g_row : nullptr);
    dsim_fwd_external(c, ex);
    if constexpr (DsimWaveDyn<Ctx, Exec>::value) {
        if (update_mass) dsim_fwd_mass(c, ex);   // H depends on the kinematics only
        if constexpr (!DsimWaveDyn<Ctx, Exec>::sums_inside) dsim_fwd_ftot(c, ex);
        dsim_fwd_dynamics_wave(c, ex, g_row, g_hinv, update_mass);
        return;
    }
    if constexpr (DsimWideDyn<Ctx, Exec>::value) {
        if (update_mass) dsim_fwd_mass(c, ex);   // H depends on the kinematics only
        dsim_fwd_dynamics_wide(c, ex, g_row, g_hinv, update_mass);
        return;
    }
    dsim_fwd_tau(c, ex);
    if (update_mass) dsim_fwd_mass(c, ex);
    dsim_fwd_solve(c, ex);
    if constexpr (wide) {
        // The first wavefront integrates (link lanes); the others copy the rest of the checkpoint row meanwhile -- X_sc .. qdd,
        // which the integrator does not touch; the head (q, qd) went out beside the kinematics -- and, on a refresh, the inverse.
        ex.mark(6);
        constexpr int NLR = Exec::NL - DSIM_NL;
        ex.fork_wave0([&](int lane) { dsim_integrate_lane(c, ex, lane); }, [&](int lane) {
            if (g_row) {
                dsim_ckpt_store_row<Ctx, NLR, 2>(c, lane, g_row);
                if (update_mass && g_hinv)
                    for (int k = lane; k < c.d.nd * c.d.nd; k += NLR) g_hinv[k] = WF(hinv)[k];
            }
        });
        return;
    }
    if (g_row) {
        // Checkpoint row = the saved block.  Global stores to this environment's private row that nobody in this launch
        // reads back: no wait -- they drain while the step goes on (16 bytes per lane and instruction: the saved block
        // and a checkpoint row are 16-byte aligned multiples of 4 words).
        ex.fire([&](int lane) {
            dsim_ckpt_store_row<Ctx, Exec::NL>(c, lane, g_row);
            if (update_mass && g_hinv)
                for (int k = lane; k < c.d.nd * c.d.nd; k += Exec::NL) g_hinv[k] = WF(hinv)[k];
        });
    }
    dsim_fwd_integrate(c, ex);
}

// ------------------------------------------------------------------------------------------------
// one env.step(): `substeps` substeps with act / muscle activations held fixed (sim.py:2104-2116).
// g_* are this environment's rows of the caller's tensors (global memory); ckpt may be null.
template <class Ctx, class Exec>
DSIM_FN void dsim_sim_step_forward(const Ctx& c, Exec& ex, int substeps, int mm_freq, const float* g_q,
                                   const float* g_qd, const float* g_act, const float* g_mact, float* g_q_out,
                                   float* g_qd_out, float* g_ckpt, int* g_status = nullptr, int env = 0) {
    const int nq = c.d.nq, nd = c.d.nd, M = c.d.M;
    ex.begin_request();
    ex.begin();
    dsim_init_static(c, ex);
    ex.run_both([&](int lane) { dsim_topo_init<false>(c, ex, lane); });   // every wave keeps its own topology records
    ex.run([&](int lane) {
        for (int k = lane; k < nq; k += Exec::NL) WF(q)[k] = g_q[k];
        for (int k = lane; k < nd; k += Exec::NL) {
            WF(qd)[k] = g_qd[k];
            WF(act)[k] = g_act[k];
        }
        for (int k = lane; k < M; k += Exec::NL) WF(mact)[k] = g_mact[k];
    });
    dsim_check_unit_quats(c, ex, g_status, env);
    for (int s = 0; s < substeps; ++s)
        dsim_fwd_substep(c, ex, (s % mm_freq) == 0, g_ckpt ? g_ckpt + (size_t)s * dsim_row(c) : nullptr,
                         g_ckpt ? dsim_ckpt_hinv(c, g_ckpt, substeps, s / mm_freq) : nullptr);
    ex.run([&](int lane) {
        for (int k = lane; k < nq; k += Exec::NL) g_q_out[k] = WF(q)[k];
        for (int k = lane; k < nd; k += Exec::NL) g_qd_out[k] = WF(qd)[k];
    });
}

// Derived body transforms of a given joint state (dsim_body_transforms, include/dsim.h): what the reference's State carries as
// body_X_sc / body_X_sm (model.py:338-392, filled by eval_rigid_fk, sim.py:1638-1678).  The kinematics phase of the step
// kernels, run once on q with qd = 0, then X_sc [L][7] and X_sm = X_sc o X_cm (joint_X_cm has an identity rotation,
// model.py:1745-1747: p_sm = p_sc + R_sc com, r_sm = r_sc) go to global memory.
template <class Ctx, class Exec>
DSIM_FN void dsim_body_transforms_only(const Ctx& c, Exec& ex, const float* g_q, float* g_xsc, float* g_xsm) {
    ex.begin_request();
    ex.begin();
    dsim_init_static(c, ex);
    ex.run_both([&](int lane) { dsim_topo_init<false>(c, ex, lane); });
    ex.run([&](int lane) {
        for (int k = lane; k < c.d.nq; k += Exec::NL) WF(q)[k] = g_q[k];
        for (int k = lane; k < c.d.nd; k += Exec::NL) WF(qd)[k] = 0.f;
    });
    dsim_fwd_kinematics(c, ex);
    ex.run([&](int lane) {
        for (int it = lane; it < 7 * c.d.L; it += Exec::NL) g_xsc[it] = WF(xsc)[it];
        if (g_xsm) {
            for (int i = lane; i < c.d.L; i += Exec::NL) {
                const v3 p = ld3(WF(xsc) + 7 * i);
                const q4 r = ldq(WF(xsc) + 7 * i + 3);
                const v3 pm = rotate(r, ld3(CF(com) + 3 * i)) + p;
                float* o = g_xsm + 7 * i;
                o[0] = pm.x; o[1] = pm.y; o[2] = pm.z; o[3] = r.x; o[4] = r.y; o[5] = r.z; o[6] = r.w;
            }
        }
    });
}

// ================================================================================================
// adjoint (hand-derived reverse sweep of one substep)
// ================================================================================================
// Preconditions: LDS q/qd hold the substep's INPUT state; kinematics, external, tau and solve have
// been recomputed for it (and dsim_fwd_composite on an update substep); hinv is the inverse that the
// forward pass used for this substep; aqn/aqdn hold the cotangents of the substep outputs (the layout aliases them with
// aq/aqd: the integrate^T phase replaces them in place, every lane reading its own link's words before writing them).
// Postconditions: aq/aqd = cotangents of the substep inputs; aact/amact/aH accumulated.

// integrate^T, solve^T (matnn.h:310-336), tau^T
// Cotangent of the mass matrix, accumulated over the substeps of one mass-matrix group.  In the specialised kernels
// every lane keeps its nd*nd/64 entries in REGISTERS across the group's substeps (lane-private accumulation: entry
// `lane + 64 m` never leaves its lane) and writes them to LDS once, at the refresh substep, for the mass-matrix adjoint.
// The LDS read-modify-write it replaces was a chain of dependent round trips in every substep.
#define DSIM_HACC_MAX 12
#define DSIM_HPF_MAX 8    // registers per lane of the inverse requested ahead of the adjoint's first substep (Exec::hpf)
template <class Ctx> struct DsimHaccRegs {
    static constexpr bool value = DsimIsStatic<Ctx>::value;
};
template <class Ctx, class Exec> DSIM_FN void dsim_hacc_zero(const Ctx& c, Exec& ex, int lane) {
    if constexpr (DsimIsStatic<Ctx>::value) {
        constexpr int NN = decltype(c.d)::nd * decltype(c.d)::nd, ACC = (NN + Exec::NL - 1) / Exec::NL;
        if constexpr (ACC <= DSIM_HACC_MAX) {
            float* acc = ex.hacc(lane);
#pragma unroll
            for (int m = 0; m < ACC; ++m) acc[m] = 0.f;
        }
    }
}

// Joint-space adjoint of the models whose links other than the free root have one hinge / prismatic dof each (DsimDims::JW_OK,
// JW_FREE_ROOT: Ant; Humanoid, Hopper and HalfCheetah qualify structurally and measured no gain), up to DSIM_JW_ND_MAX dofs: integrate^T, adj tau = H^-1 adj qdd and the per-dof cotangents of
// tau as ONE phase on the dof lanes.  A hinge dof's share of integrate^T is two loads and two multiply-adds, done by its own dof
// lane; the free root's quaternion update^T runs on lane 0 as before and its six results reach the root's dof lanes by
// v_readlane; adj qdd then travels by v_readlane into the rows of H^-1 each dof lane holds in registers (the mirror of the
// forward's qdd = H^-1 tau), and adj tau stays in the register the per-dof block needs it in.  Two phase boundaries and the LDS
// round trips of adj qdd and adj tau leave the critical path; same operations in the same order as the three phases.
#ifndef DSIM_JW_ND_MAX
#define DSIM_JW_ND_MAX 16
#endif
template <class Ctx, class Exec> struct DsimJointWave {
    static constexpr bool value = []() {
#ifdef DSIM_NO_JOINT_WAVE   // (A/B builds)
        return false;
#else
        if constexpr (std::is_empty<decltype(Ctx::d)>::value && Exec::WAVE_OPS) {
            using D = decltype(Ctx::d);
            // measured (tools/ab_min.py, adjoint launch at 1024 environments): Ant -3.9 %; Humanoid (27 dofs: 27 rows of registers and
            // broadcasts) +0.3 %, Hopper / HalfCheetah (no root chain to hide the per-dof work behind) +0.5 .. 1 % -- so: free root, <= 16 dofs
            // (all one-wave kernels of the model, helper or not: launches of any size give the same bits)
            return D::JW_OK != 0 && D::JW_FREE_ROOT != 0 && D::nd <= DSIM_JW_ND_MAX && D::nd <= Exec::NL && D::L <= Exec::NL;
        } else {
            return false;
        }
#endif
    }();
};
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_joint_wave(const Ctx& c, Exec& ex) {
    using D = decltype(c.d);
    constexpr int nd = D::nd;
    ex.run([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        const float h = c.h;
        const DsimTopoRegs& tp = ex.topo(lane);
        const bool is_dof = lane < nd;
        // ---- loads: dof role
        int type = tp.dof_type;
        DSIM_OPAQUE(type);
        const int i = tp.dof_link, cs = tp.dof_cs, d = is_dof ? lane : 0;
        const bool hinge = type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE;
        const int qi = hinge ? cs : 0;
        float hrow[nd];
#pragma unroll
        for (int j = 0; j < nd; ++j) hrow[j] = WF(hinv)[d * nd + j];
        const sv6 ft = ldsv(WF(ftot) + 6 * i);
        const float q = WF(q)[qi];
        const float tke = CF(tke)[i], tkd = CF(tkd)[i], lke = CF(lke)[i], lkd = CF(lkd)[i];
        const float lower = CF(lower)[qi], upper = CF(upper)[qi];
        const float g_q = WF(aq)[qi], g_qd_in = WF(aqd)[d], g_act = WF(aact)[d];
        // ---- integrate^T: the free root on lane 0 (its six new adj qd values in r6), a hinge on its dof lane
        float r6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (D::JW_FREE_ROOT != 0) {
            int ltype = tp.own_type;
            DSIM_OPAQUE(ltype);
            if (lane < D::L && ltype == DSIM_JOINT_FREE) {
                const int lcs = tp.own_cs, lds_ = tp.own_ds;
                const float *qq = WF(q) + lcs, *qd = WF(qd) + lds_, *qdd = WF(qdd) + lds_;
                float *aq = WF(aq) + lcs;
                const float* aqd = WF(aqd) + lds_;
                const v3 p = ld3(qq), g_pn = ld3(aq);
                const q4 r = ldq(qq + 3), g_rn = ldq(aq + 3);
                const v3 w = ld3(qd) + ld3(qdd) * h;
                const v3 gqd_w = ld3(aqd), gqd_v = ld3(aqd + 3);
                const q4 rt = r + qmul_v(w, r) * 0.5f * h;   // W = (w, 0); the integrator's own expression, rounding included
                q4 g_rt = mkq(0.f, 0.f, 0.f, 0.f);
                float il = 0.f;
                if constexpr (DsimSavedIl<Ctx, Exec>::value) {
                    il = WF(qil)[0];   // from the forward launch, through the checkpoint
                } else {
                    il = dsim_inv_len(qdot(rt, rt));
                }
                if (il > 0.0f) {
                    const q4 rn = rt * il;
                    g_rt = (g_rn + rn * (-qdot(rn, g_rn))) * il;
                }
                const q4 g_r = g_rt + qmul_v(mk3(-w.x, -w.y, -w.z), g_rt) * (0.5f * h);   // conj(W) (x) g_rt
                const q4 g_W = qmul_adj_a(r, g_rt) * (0.5f * h);
                v3 g_w = qvec(g_W) + gqd_w;
                const v3 g_dp = g_pn * h;
                const v3 g_v = g_dp + gqd_v;
                g_w += cross(p, g_dp);
                st3(aq, g_pn - cross(w, g_dp));
                stq(aq + 3, g_r);
                r6[0] = g_w.x; r6[1] = g_w.y; r6[2] = g_w.z;
                r6[3] = g_v.x; r6[4] = g_v.y; r6[5] = g_v.z;
            }
        }
        float g = g_qd_in + h * g_q;   // hinge: adj qd after integrate^T
        if constexpr (D::JW_FREE_ROOT != 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float rk = ex.bcast(r6[k], 0);
                if (lane == k) g = rk;
            }
        }
        const float aqdd = h * g;
        // ---- adj tau = H^-1 adj qdd (hinv is symmetric): adj qdd_j travels by v_readlane
        float at = 0.f;
#pragma unroll
        for (int j = 0; j < nd; ++j) at += hrow[j] * ex.bcast(aqdd, j);
        // ---- per-dof cotangents of tau
        if (is_dof) {
            WF(atau)[lane] = at;
            stsv(WF(aS) + 6 * lane, ft * (-at));
            if (hinge) {
                float dq = -tke;
                if (q < lower) dq = -tke - lke;
                if (q > upper) dq = -tke - lke;
                WF(aq)[qi] = g_q + dq * at;
                WF(aqd)[lane] = g + (-tkd - lkd) * at;
                WF(aact)[lane] = g_act + at;
            } else {
                WF(aqd)[lane] = g;   // (free root: tau has no joint-space terms)
            }
        }
    });
}

template <class Ctx, class Exec> DSIM_FN void dsim_bwd_joint_space(const Ctx& c, Exec& ex, bool update_mass) {
    ex.mark(7);
    const int nd = c.d.nd;
    constexpr bool joint_wave = DsimJointWave<Ctx, Exec>::value;
    if constexpr (joint_wave) dsim_bwd_joint_wave(c, ex);
    // integrate^T per link: loads, arithmetic, stores (aq / aqd alias aqn / aqdn: every lane replaces its own link's words)
    if constexpr (!joint_wave) ex.run([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        constexpr int NQ = dsim_mask_nq(MASK), NDF = dsim_mask_nd(MASK);
        const float h = c.h;
        for (int i = lane; i < c.d.L; i += Exec::NL) {
            int type, cs, ds;
            if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
                const DsimTopoRegs& tp = ex.topo(lane);
                type = tp.own_type; cs = tp.own_cs; ds = tp.own_ds;
                DSIM_OPAQUE(type);
            } else {
                type = CI(jtype)[i]; cs = CI(qstart)[i]; ds = CI(qdstart)[i];
            }
            const float *q = WF(q), *qd = WF(qd), *qdd = WF(qdd);
            float *aq = WF(aq), *aqd = WF(aqd), *aqdd = WF(aqdd);
            float qv[NQ > 0 ? NQ : 1], gqn[NQ > 0 ? NQ : 1], qdv[NDF > 0 ? NDF : 1], av[NDF > 0 ? NDF : 1], gqdn[NDF > 0 ? NDF : 1];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                qv[k] = q[cs + k];
                gqn[k] = aq[cs + k];
            }
#pragma unroll
            for (int k = 0; k < NDF; ++k) {
                qdv[k] = qd[ds + k];
                av[k] = qdd[ds + k];
                gqdn[k] = aqd[ds + k];
            }
            if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
                const float g = gqdn[0] + h * gqn[0];
                aqd[ds] = g;
                aqdd[ds] = h * g;
            }
            if constexpr ((MASK & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) != 0) {
                if (type == DSIM_JOINT_BALL || type == DSIM_JOINT_FREE) {
                    const bool fr = type == DSIM_JOINT_FREE;
                    const v3 w = mk3(qdv[0], qdv[1], qdv[2]) + mk3(av[0], av[1], av[2]) * h;
                    q4 r, g_rn;
                    if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                        r = fr ? mkq(qv[3], qv[4], qv[5], qv[6]) : mkq(qv[0], qv[1], qv[2], qv[3]);
                        g_rn = fr ? mkq(gqn[3], gqn[4], gqn[5], gqn[6]) : mkq(gqn[0], gqn[1], gqn[2], gqn[3]);
                    } else {
                        r = mkq(qv[0], qv[1], qv[2], qv[3]);
                        g_rn = mkq(gqn[0], gqn[1], gqn[2], gqn[3]);
                    }
                    const q4 rt = r + qmul_v(w, r) * 0.5f * h;   // W = (w, 0); the integrator's own expression, rounding included
                    q4 g_rt = mkq(0.f, 0.f, 0.f, 0.f);
                    float il = 0.f;
                    if constexpr (DsimSavedIl<Ctx, Exec>::value) {
                        il = WF(qil)[0];   // from the forward launch, through the checkpoint: no square root, no division here
                    } else {
                        il = dsim_inv_len(qdot(rt, rt));
                    }
                    if (il > 0.0f) {
                        const q4 rn = rt * il;
                        g_rt = (g_rn + rn * (-qdot(rn, g_rn))) * il;
                    }
                    const q4 g_r = g_rt + qmul_v(mk3(-w.x, -w.y, -w.z), g_rt) * (0.5f * h);   // conj(W) (x) g_rt
                    const q4 g_W = qmul_adj_a(r, g_rt) * (0.5f * h);
                    v3 g_w = qvec(g_W) + mk3(gqdn[0], gqdn[1], gqdn[2]);
                    if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                        if (fr) {
                            const v3 p = mk3(qv[0], qv[1], qv[2]);
                            const v3 g_pn = mk3(gqn[0], gqn[1], gqn[2]);
                            const v3 g_dp = g_pn * h;
                            const v3 g_v = g_dp + mk3(gqdn[3], gqdn[4], gqdn[5]);
                            g_w += cross(p, g_dp);
                            st3(aq + cs, g_pn - cross(w, g_dp));
                            st3(aqd + ds + 3, g_v);
                            st3(aqdd + ds + 3, g_v * h);
                        }
                    }
                    stq(aq + cs + (fr ? 3 : 0), g_r);
                    st3(aqd + ds, g_w);
                    st3(aqdd + ds, g_w * h);
                }
            }
        }
    });
    if constexpr (!joint_wave) ex.run([&](int lane) {
        for (int i = lane; i < nd; i += Exec::NL) {
            WF(atau)[i] = dsim_dot_n(WF(hinv) + i * nd, WF(aqdd), nd);  // hinv is symmetric
        }
    });
    // Three blocks that touch disjoint LDS words: the mass-matrix cotangent accumulators, the per-dof cotangents of tau, and
    // af -- the last one is handed to the helper wavefront where there is one (Exec::fork_join)
    auto tau_adjoint_per_dof = [&](int lane) __attribute__((always_inline)) {
        for (int d = lane; d < nd; d += Exec::NL) {
            int i, type, cs, ds;
            if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
                const DsimTopoRegs& tp = ex.topo(lane);
                i = tp.dof_link; type = tp.dof_type; cs = tp.dof_cs; ds = tp.dof_ds;
                DSIM_OPAQUE(type);
            } else {
                i = CI(dof_link)[d]; type = CI(jtype)[i]; cs = CI(qstart)[i]; ds = CI(qdstart)[i];
            }
            // loads first (aq / aqd / aact are read-modify-write), stores last
            const float at = WF(atau)[d];
            const sv6 ft = ldsv(WF(ftot) + 6 * i);
            const bool hinge = type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE;
            const bool ball = type == DSIM_JOINT_BALL;
            const int qi = hinge ? cs : (ball ? cs + (d - ds) : 0);
            const float q = WF(q)[qi];
            const float tke = CF(tke)[i], tkd = CF(tkd)[i], lke = CF(lke)[i], lkd = CF(lkd)[i];
            const float lower = CF(lower)[qi], upper = CF(upper)[qi];
            const float g_q = WF(aq)[qi], g_qd = WF(aqd)[d], g_act = WF(aact)[d];
            stsv(WF(aS) + 6 * d, ft * (-at));
            if (hinge) {
                float dq = -tke;
                if (q < lower) dq = -tke - lke;
                if (q > upper) dq = -tke - lke;
                WF(aq)[qi] = g_q + dq * at;
                WF(aqd)[d] = g_qd + (-tkd - lkd) * at;
                WF(aact)[d] = g_act + at;
            } else if (ball) {
                WF(aq)[qi] = g_q + (-tke) * at;
                WF(aqd)[d] = g_qd + (-tkd) * at;
            }
        }
    };
    // (main) the mass-matrix cotangent accumulators (registers of the main wave) and the per-dof block, (side) af
    auto af_block = [&](int lane) __attribute__((always_inline)) {
        if constexpr (DsimTrunk<Ctx, Exec>::value) {
            // Trunk decomposition of the ancestor sums: u_j = sum over the own dofs of link j of S_d atau_d; a trunk link's
            // prefix P_t = P_parent + u_t is evaluated by lanes 0..5 straight from S and atau (the trunk is a handful of
            // links whose dofs are compile-time constants); a light link gets af_j = -(P of its nearest trunk ancestor +
            // the u rows of its light ancestors-or-self), at most LCAP of them.  The u rows live in avtot, which is dead here.
            using D = decltype(c.d);
            const DsimTopoRegs& tp = ex.topo(lane);
            constexpr int PASSES = (6 * D::NLT + Exec::NL - 1) / Exec::NL;
            float* urow = WF(avtot);
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = tp.tl_row[p], d = tp.tu_d[p];
                const int dd = d < 0 ? 0 : d;
                const float sv = WF(S)[6 * dd + tp.tl_k[p]], tv = WF(atau)[dd];
                if (row >= 0) urow[row] = d < 0 ? 0.f : sv * tv;
            }
            if (lane < 6) {
                float P[D::NT];
                dsim_static_for<0, D::NT>([&](auto uu) {
                    constexpr int u = decltype(uu)::value;
                    float acc = D::tr_par[u] >= 0 ? P[D::tr_par[u] >= 0 ? D::tr_par[u] : 0] : 0.f;
#pragma unroll
                    for (int e = 0; e < D::tr_nd[u]; ++e) acc += WF(S)[6 * (D::tr_d0[u] + e) + lane] * WF(atau)[D::tr_d0[u] + e];
                    P[u] = acc;
                    WF(af)[6 * D::trunk[u] + lane] = 0.f - acc;
                });
            }
            ex.lds_fence();
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int row = tp.tl_row[p], top = tp.ta_top[p];
                int m = tp.ta_m[p];
                DSIM_OPAQUE(m);
                float x[D::LCAP];
#pragma unroll
                for (int e = 0; e < D::LCAP; ++e) x[e] = urow[top + 6 * e];
                float acc = WF(af)[tp.ta_tp[p]];
#pragma unroll
                for (int e = 0; e < D::LCAP; ++e) acc -= ((m >> e) & 1) ? x[e] : 0.f;
                if (row >= 0) WF(af)[row] = acc;
            }
        }
        for (int it = Exec::NL - 1 - lane; it < (DsimTrunk<Ctx, Exec>::value ? 0 : 6 * c.d.L); it += Exec::NL) {
            const int j = it / 6, k = it - 6 * j;
            float acc = 0.f;
            if constexpr (DsimAdofRegs<Ctx, Exec::NL>::value) {
                // the dof list of this lane's item is in registers: operands in ONE round trip
                constexpr int B = decltype(c.d)::nd;
                const DsimTopoRegs& tp = ex.topo(lane);
                float sv[B], tv[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    sv[u] = WF(S)[6 * tp.adof[u] + k];
                    tv[u] = WF(atau)[tp.adof[u]];
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const float term = sv[u] * tv[u];   // the zero dof contributes (finite S word) * 0
                    acc -= term;
                }
                WF(af)[it] = acc;
                continue;
            }
            const dsim_int_a* lst = CI(adof_list);
            int e = CI(adof_start)[j];
            const int e1 = CI(adof_start)[j + 1];
            if constexpr (DsimIsStatic<Ctx>::value) {
                // bounded list: indices in one round trip, operands in a second one (see dsim_range_sum_b)
                constexpr int B = decltype(c.d)::nd <= 16 ? decltype(c.d)::nd : 8;
                const int cnt = e1 - e;
                int dd[B];
                float sv[B], tv[B];
#pragma unroll
                for (int u = 0; u < B; ++u) dd[u] = lst[e + (u < cnt ? u : 0)];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    sv[u] = WF(S)[6 * dd[u] + k];
                    tv[u] = WF(atau)[dd[u]];
                }
#pragma unroll
                for (int u = 0; u < B; ++u) acc -= (u < cnt) ? sv[u] * tv[u] : 0.f;
                e += cnt < B ? cnt : B;
            }
            for (; e + 4 <= e1; e += 4) {
                const int d0 = lst[e], d1 = lst[e + 1], d2 = lst[e + 2], d3 = lst[e + 3];
                const float s0 = WF(S)[6 * d0 + k], s1 = WF(S)[6 * d1 + k], s2 = WF(S)[6 * d2 + k], s3 = WF(S)[6 * d3 + k];
                const float t0 = WF(atau)[d0], t1 = WF(atau)[d1], t2 = WF(atau)[d2], t3 = WF(atau)[d3];
                acc -= s0 * t0;
                acc -= s1 * t1;
                acc -= s2 * t2;
                acc -= s3 * t3;
            }
            for (; e < e1; ++e) acc -= WF(S)[6 * lst[e] + k] * WF(atau)[lst[e]];
            WF(af)[it] = acc;
        }
    };
    ex.fork_join([&](int lane) {
        bool in_regs = false;
        if constexpr (DsimIsStatic<Ctx>::value) {
            constexpr int NN = decltype(c.d)::nd * decltype(c.d)::nd, ACC = (NN + Exec::NL - 1) / Exec::NL;
            if constexpr (ACC <= DSIM_HACC_MAX) {
                in_regs = true;
                float* acc = ex.hacc(lane);
#pragma unroll
                for (int m = 0; m < ACC; ++m) {
                    const int it = lane + Exec::NL * m;
                    if (it < NN) {
                        const int i = it / decltype(c.d)::nd, j = it - decltype(c.d)::nd * i;
                        acc[m] -= WF(atau)[i] * WF(qdd)[j];
                        if (update_mass) WF(aH)[it] = acc[m];
                    }
                }
            }
        }
        if (!in_regs) {
            for (int it = lane; it < nd * nd; it += Exec::NL) {
                const int i = it / nd, j = it - nd * i;
                WF(aH)[it] -= WF(atau)[i] * WF(qdd)[j];
            }
        }
        // cotangent of body_f_s[j]: f_tot of every ancestor-or-self i of j contains f[j] and tau_d = -S_d . f_tot[link(d)], so
        // af[j] = -sum over the dofs d of all ancestors-or-self of j of S_d atau_d (same phase: reads only S and atau;
        // items are dealt from the top lane down so that they do not pile onto the lanes of the per-dof loop above)
        if constexpr (!joint_wave) tau_adjoint_per_dof(lane);
    }, af_block);

}

// contacts^T and muscles^T.  The cotangent with respect to the POSE of a body is kept as a world-frame wrench (torque
// about the origin, force): a cotangent a_x of a body-fixed point at world position x is the wrench (x x a_x, a_x), and
// wrenches of different points / different sources simply add (dsim_bwd_bodies).  Per contact: [wrench 6, cotangent of
// the body's twist 6]; per muscle segment: [wrench on link 0, wrench on link 1, cotangent of the activation].
// Runs inside the first body-level phase (both only need af); items are dealt from the top lane of the wavefront down.
// NLX: lanes the items are dealt to (Exec::NL: all lanes of the environment, from the top lane down; fewer -- the wavefronts behind
// the first one, DsimWideOverlap: contacts from the top lane down, segments from lane 0 up, so that a lane's second item is of
// the other kind where the lists allow it)
template <int NLX, class Ctx, class Exec> DSIM_FN void dsim_bwd_external_items_n(const Ctx& c, Exec& ex, int real_lane) {
    constexpr bool ALL = NLX == Exec::NL;
    const int lane = NLX - 1 - real_lane;
    for (int k = lane; k < c.d.C; k += NLX) {
        int b;
        if constexpr (ALL && DsimContactRegs<Ctx, Exec::NL>::value) b = ex.topo(real_lane).cbody_b;
        else b = CI(cbody)[k];
        const v3 xp = ld3(WF(xsc) + 7 * b);
        const q4 xq = ldq(WF(xsc) + 7 * b + 3);
        const sv6 vb = ldsv(WF(v) + 6 * b);
        const sv6 A = ldsv(WF(af) + 6 * b);  // cotangent of body_f_s[b]
        const float* mat = CF(cmat) + 4 * k;
        const float ke = mat[0], kd = mat[1], kf = mat[2], mu = mat[3];
        const v3 cp = ld3(CF(cpoint) + 3 * k);
        const float cdist = CF(cdist)[k];
        const v3 x = xp + rotate(xq, cp);  // world position of the body-fixed contact point
        v3 p = x;
        p.y -= cdist;
        const float cc = p.y;
        sv6 wr = zerosv(), tw = zerosv();
        if (cc < 0.0f) {
            const v3 dpdt = vb.v + cross(vb.w, p);
            const float vn = dpdt.y;
            const v3 vt = mk3(dpdt.x, 0.f, dpdt.z);
            const float fn = cc * ke;
            const float vmin = vn < 0.0f ? vn : 0.0f;
            const float fd = vmin * kd * (0.0f - cc);
            const float vt2 = dot(vt, vt), ilt = dsim_inv_len(vt2), lt = vt2 * ilt;   // (as in dsim_contact_wrench)
            const float a1 = kf * lt, a2 = 0.0f - mu * cc * ke;
            const bool first = a1 < a2;
            const float smin = first ? a1 : a2;
            const v3 nhat = vt * ilt;
            const v3 ft = nhat * smin;
            const v3 F = mk3(ft.x, fn + fd, ft.z);
            v3 a_p = cross(F, A.w);
            const v3 a_F = A.v + cross(A.w, p);
            const float a_fnfd = a_F.y;
            const v3 a_ft = a_F;  // ft.y == 0 and receives a_F.y too, but vt.y has no dependence (see below)
            const float a_s = dot(nhat, a_ft);
            const v3 a_nhat = a_ft * smin;
            float a_lt = 0.f, a_c = 0.f, a_vn = 0.f;
            if (first) a_lt += kf * a_s;
            else a_c += -mu * ke * a_s;
            const v3 a_vt = (a_nhat - nhat * dot(nhat, a_nhat)) * ilt + nhat * a_lt;   // (zero at vt = 0: normalize / length have zero gradients there)
            if (vn < 0.0f) a_vn += kd * (0.0f - cc) * a_fnfd;
            a_c += -vmin * kd * a_fnfd;
            a_c += ke * a_fnfd;
            // vt = dpdt - n*vn ; vn = n.dpdt  (n = +y)
            v3 a_dp = a_vt;
            a_vn += -a_vt.y;
            a_dp.y += a_vn;
            a_p.y += a_c;
            // dpdt = v + w x p
            a_p += cross(a_dp, vb.w);
            wr = mksv(cross(x, a_p), a_p);
            tw = mksv(cross(p, a_dp), a_dp);
        }
        float* o = WF(acx) + 12 * k;
        stsv(o, wr);
        stsv(o + 6, tw);
    }
    for (int s = ALL ? lane : real_lane; s < c.d.NS; s += NLX) {
        const DsimSegRec sg = dsim_seg_rec(c, s);
        const DsimSegIn in = dsim_seg_load(c, sg);
        // (af rows are [L][6], X_sc rows [L][7]: 6 * link = offset - offset / 7)
        const sv6 A0 = ldsv(WF(af) + (sg.x0 - sg.x0 / 7)), A1 = ldsv(WF(af) + (sg.x1 - sg.x1 / 7));
        const v3 pos0 = in.p0 + rotate(in.r0, in.m0), pos1 = in.p1 + rotate(in.r1, in.m1);
        const float act = in.act;
        const v3 d = pos1 - pos0;
        const float il = dsim_inv_len_item(dot(d, d));
        const v3 n = d * il, f = n * act;   // (n = 0 for d = 0: every term below is zero then)
        const v3 a_f = cross(A1.w, pos1) + A1.v - cross(A0.w, pos0) - A0.v;
        v3 a_p0 = -cross(f, A0.w);
        v3 a_p1 = cross(f, A1.w);
        const float a_act = dot(n, a_f);
        const v3 a_n = a_f * act;
        const v3 a_d = (a_n - n * dot(n, a_n)) * il;
        a_p1 += a_d;
        a_p0 -= a_d;
        // the forward wrench rows are dead by now: same buffer, same body-sorted rows + one activation cotangent per segment
        stsv(WF(mus) + sg.r0, mksv(cross(pos0, a_p0), a_p0));
        stsv(WF(mus) + sg.r1, mksv(cross(pos1, a_p1), a_p1));
        WF(mus)[dsim_mus_act(c) + s] = a_act;
    }
}
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_external_items(const Ctx& c, Exec& ex, int real_lane) {
    dsim_bwd_external_items_n<Exec::NL>(c, ex, real_lane);
}

// mass matrix^T (update substeps): aH -> aS (added), ai10m.
// H[a][b] = S_a^T Ic[deeper link] S_b for related dofs.  For dof a on link la the related dofs are (i) the dofs of the
// strict subtree of la -- the deeper link is theirs, the term is S_a . F_b with F_b = Ic[link(b)] S_b -- and (ii) the
// dofs of the ancestors-or-self of la (the list adof(la)) -- the deeper link is la, the term is S_a . Ic[la] S_b.
// Both sets are short lists (a contiguous dof range with pre-order numbering), walked two entries per LDS round trip.
// The same for the specialised kernels of pre-order trees: both lists of a dof are BOUNDED at compile time (DsimDims::SDMAX,
// ADMAX), so their entries are fetched in batches -- the strict-subtree dofs are a contiguous range (no index loads at all), the
// ancestor dofs one batch of indices, one of operands -- and masked with 0 / 1 weights, instead of two run-time loops of
// dependent loads, two entries per round trip (measured on the old form: Ant 7.9 k, SNUHumanoid 12.8 k cycles per refresh, the
// latter six times per env-step).  Same terms in the same order: bit-identical to the loops.
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_mass_bounded(const Ctx& c, Exec& ex) {
    using D = decltype(c.d);
    // (batches of CH entries: all of a long list at once costs Humanoid > 256 VGPRs)
    constexpr int nd = D::nd, CH = 4, B1 = (D::SDMAX + CH - 1) / CH * CH, B2 = (D::ADMAX + CH - 1) / CH * CH;
    // the two lists are independent blocks (aS / aic10): the second goes to the helper wavefront, or to the upper half of a
    // workgroup of several waves
    constexpr int OFF = (!Exec::HAS_HELPER && Exec::NL >= 128) ? Exec::NL / 2 : 0;
    ex.fork_join([&](int lane) {
        const float* aH = WF(aH);
        for (int a = lane; a < nd; a += Exec::NL) {
            const int la = CI(dof_link)[a];
            const int nsub = CI(linfo)[8 * la + 5], e0 = CI(adof_start)[la], cnt = CI(adof_start)[la + 1] - e0;
            const int b0 = CI(qdstart)[la + 1], nb = CI(qdstart)[la + nsub] - b0;
            sv6 acc = zerosv(), u = zerosv();
#pragma unroll 1
            for (int c0 = 0; c0 < B1; c0 += CH) {
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const bool in = c0 + e < nb;
                    const int b = in ? b0 + c0 + e : a;   // (past the end: an in-range row, weight 0)
                    const float w = aH[a * nd + b] + aH[b * nd + a];
                    acc += ldsv(WF(F) + 6 * b) * (in ? w : 0.f);
                }
            }
#pragma unroll 1
            for (int c0 = 0; c0 < B2; c0 += CH) {
                int dd[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) dd[e] = CI(adof_list)[e0 + (c0 + e < cnt ? c0 + e : 0)];
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int d0 = dd[e];
                    const float w = aH[a * nd + d0] + aH[d0 * nd + a];
                    u += ldsv(WF(S) + 6 * d0) * ((c0 + e < cnt) ? w : 0.f);
                }
            }
            acc += inertia_mul(ld_i10(WF(ic10) + 10 * la), u);
            float* o = WF(aS) + 6 * a;
            const sv6 g = ldsv(o);
            stsv(o, g + acc);
        }
    }, [&](int lane) {
        const float* aH = WF(aH);
        for (int t = lane; t < nd + OFF; t += Exec::NL) {
            const int b = t - OFF;
            if (b < 0) continue;
            float g[10];
            for (int k = 0; k < 10; ++k) g[k] = 0.f;
            const int j = CI(dof_link)[b];
            const sv6 Sb = ldsv(WF(S) + 6 * b);
            const int e0 = CI(adof_start)[j], cnt = CI(adof_start)[j + 1] - e0;
#pragma unroll 1
            for (int c0 = 0; c0 < B2; c0 += CH) {
                int aa[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) aa[e] = CI(adof_list)[e0 + (c0 + e < cnt ? c0 + e : 0)];
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int a0 = aa[e];
                    const float w0 = (a0 == b) ? aH[a0 * nd + a0] : aH[a0 * nd + b] + aH[b * nd + a0];
                    inertia_bilinear_adj(g, ldsv(WF(S) + 6 * a0), Sb, (c0 + e < cnt && a0 <= b) ? w0 : 0.f);
                }
            }
            for (int k = 0; k < 10; ++k) WF(aic10)[10 * b + k] = g[k];
        }
    });
    ex.run([&](int lane) {
        for (int it = lane; it < 10 * c.d.L; it += Exec::NL) {
            const int i = it / 10, k = it - 10 * i;
            const int e0 = CI(adof_start)[i], cnt = CI(adof_start)[i + 1] - e0;
            float acc = 0.f;
#pragma unroll 1
            for (int c0 = 0; c0 < B2; c0 += CH) {
                int bb[CH];
#pragma unroll
                for (int e = 0; e < CH; ++e) bb[e] = CI(adof_list)[e0 + (c0 + e < cnt ? c0 + e : 0)];
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const float x = WF(aic10)[10 * bb[e] + k];
                    acc += (c0 + e < cnt) ? x : 0.f;
                }
            }
            WF(ai10m)[it] = acc;
        }
    });
}

template <class Ctx, class Exec> DSIM_FN void dsim_bwd_mass(const Ctx& c, Exec& ex) {
    ex.mark(9);
    if constexpr (DsimIsStatic<Ctx>::value) {
#ifndef DSIM_NO_BOUNDED_MASS   // (A/B builds)
        if constexpr ((decltype(c.d)::flags & DSIM_F_RANGES) != 0) {
            dsim_bwd_mass_bounded(c, ex);
            return;
        }
#endif
    }
    const int nd = c.d.nd;
    ex.run([&](int lane) {
        for (int a = lane; a < nd; a += Exec::NL) {
            const int la = CI(dof_link)[a];
            sv6 acc = zerosv(), u = zerosv();
            const float* aH = WF(aH);
            // pair weight: H[a][b] and H[b][a] are the same bilinear form (for a == b this gives the factor 2 of the quadratic form)
            if (c.d.flags & DSIM_F_RANGES) {
                const int nsub = reinterpret_cast<const dsim_int_a*>(c.k)[c.o.linfo + 8 * la + 5];
                int b = CI(qdstart)[la + 1];
                const int b1 = CI(qdstart)[la + nsub];
                for (; b + 2 <= b1; b += 2) {
                    const float w0 = aH[a * nd + b] + aH[b * nd + a], w1 = aH[a * nd + b + 1] + aH[(b + 1) * nd + a];
                    const sv6 f0 = ldsv(WF(F) + 6 * b), f1 = ldsv(WF(F) + 6 * b + 6);
                    acc += f0 * w0;
                    acc += f1 * w1;
                }
                if (b < b1) acc += ldsv(WF(F) + 6 * b) * (aH[a * nd + b] + aH[b * nd + a]);
                const dsim_int_a* lst = CI(adof_list);
                int e = CI(adof_start)[la];
                const int e1 = CI(adof_start)[la + 1];
                for (; e + 2 <= e1; e += 2) {
                    const int d0 = lst[e], d1 = lst[e + 1];
                    const float w0 = aH[a * nd + d0] + aH[d0 * nd + a], w1 = aH[a * nd + d1] + aH[d1 * nd + a];
                    const sv6 s0 = ldsv(WF(S) + 6 * d0), s1 = ldsv(WF(S) + 6 * d1);
                    u += s0 * w0;
                    u += s1 * w1;
                }
                if (e < e1) {
                    const int d0 = lst[e];
                    u += ldsv(WF(S) + 6 * d0) * (aH[a * nd + d0] + aH[d0 * nd + a]);
                }
            } else {
                for (int b = 0; b < nd; ++b) {
                    const int r = CI(rel)[a * nd + b];
                    if (r == 0) continue;
                    const float w = aH[a * nd + b] + aH[b * nd + a];
                    if (r == 1 && CI(dof_link)[b] != la) acc += ldsv(WF(F) + 6 * b) * w;
                    else u += ldsv(WF(S) + 6 * b) * w;
                }
            }
            acc += inertia_mul(ld_i10(WF(ic10) + 10 * la), u);
            float* o = WF(aS) + 6 * a;
            const sv6 g = ldsv(o);
            stsv(o, g + acc);
        }
        // cotangent of the composite inertia of link(b), one partial per dof b (upper lanes: they do not pile onto the lanes
        // of the aS loop above): pairs {a, b} whose deeper link is link(b), i.e. a in adof(link(b)) with a <= b (same-link
        // pairs are counted once; ancestors have smaller dof indices)
        for (int t = lane; t < nd + 32; t += Exec::NL) {
            const int b = t - 32;
            if (b < 0) continue;
            float g[10];
            for (int k = 0; k < 10; ++k) g[k] = 0.f;
            const int j = CI(dof_link)[b];
            const sv6 Sb = ldsv(WF(S) + 6 * b);
            const float* aH = WF(aH);
            const dsim_int_a* lst = CI(adof_list);
            int e = CI(adof_start)[j];
            const int e1 = CI(adof_start)[j + 1];
            for (; e + 2 <= e1; e += 2) {
                const int a0 = lst[e], a1 = lst[e + 1];
                const float w0 = (a0 == b) ? aH[a0 * nd + a0] : aH[a0 * nd + b] + aH[b * nd + a0];
                const float w1 = (a1 == b) ? aH[a1 * nd + a1] : aH[a1 * nd + b] + aH[b * nd + a1];
                const sv6 s0 = ldsv(WF(S) + 6 * a0), s1 = ldsv(WF(S) + 6 * a1);
                if (a0 <= b) inertia_bilinear_adj(g, s0, Sb, w0);
                if (a1 <= b) inertia_bilinear_adj(g, s1, Sb, w1);
            }
            if (e < e1) {
                const int a0 = lst[e];
                const float w0 = (a0 == b) ? aH[a0 * nd + a0] : aH[a0 * nd + b] + aH[b * nd + a0];
                if (a0 <= b) inertia_bilinear_adj(g, ldsv(WF(S) + 6 * a0), Sb, w0);
            }
            for (int k = 0; k < 10; ++k) WF(aic10)[10 * b + k] = g[k];
        }
    });
    // ai10m[i] = sum over the dofs b of all ancestors-or-self of link i
    ex.run([&](int lane) {
        for (int it = lane; it < 10 * c.d.L; it += Exec::NL) {
            const int i = it / 10, k = it - 10 * i;
            WF(ai10m)[it] = dsim_gather_sum(WF(aic10), 10, k, CI(adof_list), CI(adof_start)[i], CI(adof_start)[i + 1], 0.f);
        }
    });
}

// Cotangent of a link's world inertia (10 parameters, g) as a pose wrench.  A rigid perturbation (dtheta about the world
// origin, dp) of the body changes A by [dtheta]x A - A [dtheta]x + m(2 c.dp 1 - dp c^T - c dp^T) and h by dtheta x h + m dp;
// contracting with g (g[5], g[6], g[8] are the derivatives with respect to the single stored off-diagonal parameters)
// gives torque = axial(G2 A - A G2) + h x g_h, force = 2 tr(G) h - G2 h + m g_h with G2 = [[2gxx,gxy,gxz],[gxy,2gyy,gyz],[gxz,gyz,2gzz]].
DSIM_FN sv6 inertia_pose_wrench(const inertia10& I, const float* g) {
    const float gxx = g[4], gxy = g[5], gxz = g[6], gyy = g[7], gyz = g[8], gzz = g[9];
    const v3 gh = mk3(g[1], g[2], g[3]);
    // M = G2 * A, only the antisymmetric part is needed
    const float m01 = 2.f * gxx * I.axy + gxy * I.ayy + gxz * I.ayz, m10 = gxy * I.axx + 2.f * gyy * I.axy + gyz * I.axz;
    const float m02 = 2.f * gxx * I.axz + gxy * I.ayz + gxz * I.azz, m20 = gxz * I.axx + gyz * I.axy + 2.f * gzz * I.axz;
    const float m12 = gxy * I.axz + 2.f * gyy * I.ayz + gyz * I.azz, m21 = gxz * I.axy + gyz * I.ayy + 2.f * gzz * I.ayz;
    const v3 tA = mk3(m21 - m12, m02 - m20, m10 - m01);
    const float tr2 = 2.f * (gxx + gyy + gzz);
    const v3 g2h = mk3(2.f * gxx * I.h.x + gxy * I.h.y + gxz * I.h.z, gxy * I.h.x + 2.f * gyy * I.h.y + gyz * I.h.z,
                       gxz * I.h.x + gyz * I.h.y + 2.f * gzz * I.h.z);
    return mksv(tA + cross(I.h, gh), I.h * tr2 - g2h + gh * I.m);
}

// body level: f^T, velocity / acceleration recursions^T, joint motion^T, and the pose cotangents.
//
// Poses enter a substep through everything that is rigidly attached to a link: its world inertia, its centre of mass
// (gravity), its contact points / muscle waypoints, and the motion subspace of its child joints (attached to the
// link's X_sc, i.e. to the child's X_sj).  Instead of differentiating the quaternion formulas of the forward
// kinematics link by link (the reference's generated adjoint does, spatial.h / quat.h adj_* functions), every such
// cotangent is expressed as a world-frame wrench W_i on the link it is attached to; a joint coordinate q_d moves the
// whole subtree of its link rigidly with twist S_d dq_d, hence   adj q_d = S_d . sum_{i in subtree(link(d))} W_i,
// one subtree sum and one 6-dot per dof -- the same shape as tau = -S . f_tot in the forward pass.  Quaternion
// coordinates (ball, free) get the tangent-space cotangent 2 (t, 0) (x) q; the component along q itself (which the
// reference's literal differentiation also produces and the integrator's normalisation annihilates, DESIGN.md
// "Quaternion radial component") is zero here.
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_bodies(const Ctx& c, Exec& ex, bool update_mass) {
    ex.mark(10);
    // (main) per-link cotangents, (side) contacts^T / muscles^T per item: both only read af and forward quantities, and
    // write disjoint arrays -- the side block goes to the helper wavefront where there is one
    ex.fork_join([&](int lane) {
        for (int i = lane; i < c.d.L; i += Exec::NL) {
            const inertia10 I = ld_i10(WF(i10) + 10 * i);
            const sv6 v = ldsv(WF(v) + 6 * i), a = ldsv(WF(a) + 6 * i), r = ldsv(WF(af) + 6 * i);
            const v3 grav = ld3(CF(grav));
            float g[10];
            for (int k = 0; k < 10; ++k) g[k] = update_mass ? WF(ai10m)[10 * i + k] : 0.f;
            const sv6 hv = inertia_mul(I, v);
            sv6 a_v, a_hv;
            a_v.w = cross(hv.w, r.w) + cross(hv.v, r.v);
            a_v.v = cross(hv.v, r.w);
            a_hv.w = cross(r.w, v.w);
            a_hv.v = cross(r.w, v.v) + cross(r.v, v.w);
            a_v += inertia_mul(I, a_hv);
#ifdef DSIM_INJECT_ADJ_ERROR
            // developer / test builds ONLY (tests/test_probe_ledger.py, -DDSIM_INJECT_ADJ_ERROR=1.01f): a deliberate relative error
            // in one adjoint phase, to show that the parity tests' probed tolerances cannot absorb a real adjoint defect
            a_v = a_v * DSIM_INJECT_ADJ_ERROR;
#endif
            inertia_bilinear_adj(g, r, a, 1.0f);
            inertia_bilinear_adj(g, a_hv, v, 1.0f);
            // pose wrench of the link: inertia + gravity (f_g = (c x m g, m g) enters f with a minus sign)
            sv6 W = inertia_pose_wrench(I, g);
            const v3 rg = cross(r.w, grav);
            W.w += cross(I.h, rg);
            W.v += rg * I.m;
            if constexpr (!Exec::HAS_HELPER) stsv(WF(aa) + 6 * i, inertia_mul(I, r));
            stsv(WF(av) + 6 * i, a_v);  // contact cotangents are added by the item-parallel gather below
            stsv(WF(aw) + 6 * i, W);
        }
    }, [&](int lane) {
        dsim_bwd_external_items(c, ex, lane);
        if constexpr (Exec::HAS_HELPER) {   // aa = I af: evens out the two waves (the main block is the longer one)
            for (int i = lane; i < c.d.L; i += Exec::NL)
                stsv(WF(aa) + 6 * i, inertia_mul(ld_i10(WF(i10) + 10 * i), ldsv(WF(af) + 6 * i)));
        }
    });
    ex.run([&](int lane) {
        for (int m = lane; m < c.d.M; m += Exec::NL) {
            // a muscle's active segments are consecutive rows of `mus`: batched range sum, not a serial chain of loads
            const float g = WF(amact)[m];
            WF(amact)[m] = g + dsim_muscle_act_sum(c, m);
        }
        if constexpr (DsimTrunk<Ctx, Exec>::value) dsim_trunk_sum<true>(c, ex, lane, WF(aa), nullptr, 0, 0, WF(aatot));
        for (int it = lane; it < (DsimTrunk<Ctx, Exec>::value ? 0 : 6 * c.d.L); it += Exec::NL) {
            const int i = it / 6, k = it - 6 * i;
            int n_known = -1;
            if constexpr (DsimSixRegs<Ctx, Exec::NL>::value) {
                n_known = ex.topo(lane).six_n;
                DSIM_OPAQUE(n_known);
            }
            if constexpr (DsimSumMasks<Ctx, Exec::NL>::value)
                WF(aatot)[it] = dsim_range_sum_m<decltype(c.d)::L>(WF(aa), 6, k, i, ex.topo(lane).lmask, 0.f);
            else
                WF(aatot)[it] = dsim_subtree_sum(c, WF(aa), 6, k, i, n_known);
        }
        // per-body gather of the muscle pose wrenches, first step: chunk sums (contact cotangents of models without muscles
        // go straight into the subtree sums below)
        if (c.d.NS > 0) dsim_muscle_chunk_sums(c, lane, Exec::NL);
    });
    if (c.d.NS > 0) {
        ex.run([&](int lane) {
            for (int it = lane; it < 12 * c.d.L; it += Exec::NL) {
                const int i = it / 12, r = it - 12 * i;
                // r < 6: pose wrench of the body = its muscle rows (chunk sums) + its contacts; r >= 6: the twist cotangent of
                // its contacts
                float acc = 0.f;
                if (r < 6) acc = dsim_body_chunk_sum(c, i, r, 0.f);
                WF(agx)[it] = dsim_body_contact_sum(c, i, WF(acx), 12, r, acc);
            }
        });
    }
    ex.run([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        for (int i = lane; i < c.d.L; i += Exec::NL) {
            int type, ds;
            if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
                const DsimTopoRegs& tp = ex.topo(lane);
                type = tp.own_type; ds = tp.own_ds;
                DSIM_OPAQUE(type);
            } else {
                type = CI(jtype)[i]; ds = CI(qdstart)[i];
            }
            const sv6 v = ldsv(WF(v) + 6 * i), A = ldsv(WF(aatot) + 6 * i);
            sv6 a_v = ldsv(WF(av) + 6 * i), a_vj;
            if (c.d.NS > 0) a_v += ldsv(WF(agx) + 12 * i + 6);   // contact twist cotangents gathered per body
            // vj = S qd of the link's own joint (not stored by the forward pass).  Operands of the hinge and free cases are
            // fetched unconditionally, together with the loads above (one round trip; unused words are harmless)
            const sv6 S0 = ldsv(WF(S) + 6 * ds);
            const float qd0 = WF(qd)[ds];
            sv6 vj = zerosv();
            if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                const sv6 qd6 = mksv(mk3(qd0, WF(qd)[ds + 1], WF(qd)[ds + 2]), ld3(WF(qd) + ds + 3));
                if (type == DSIM_JOINT_FREE) vj = qd6;
            }
            if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) vj = S0 * qd0;
            if constexpr ((MASK & DSIM_TM(DSIM_JOINT_BALL)) != 0) {
                if (type == DSIM_JOINT_BALL)
                    for (int k = 0; k < 3; ++k) vj += ldsv(WF(S) + 6 * (ds + k)) * WF(qd)[ds + k];
            }
            a_v.w += cross(vj.w, A.w) + cross(vj.v, A.v);
            a_v.v += cross(vj.w, A.v);
            a_vj.w = cross(A.w, v.w) + cross(A.v, v.v);
            a_vj.v = cross(A.v, v.w);
            stsv(WF(av) + 6 * i, a_v);
            stsv(WF(avj) + 6 * i, a_vj);
        }
    });
    ex.run([&](int lane) {
        if constexpr (DsimTrunk<Ctx, Exec>::value) {
            dsim_trunk_sum<true>(c, ex, lane, WF(av), WF(acx), 12, 6, WF(avtot));
            return;
        }
        for (int it = lane; it < 6 * c.d.L; it += Exec::NL) {
            const int i = it / 6, k = it - 6 * i;
            int n_known = -1;
            if constexpr (DsimSixRegs<Ctx, Exec::NL>::value) n_known = ex.topo(lane).six_n;
            (void)n_known;
            WF(avtot)[it] = dsim_subtree_contact_sum(c, ex, lane, i, WF(av), k, WF(acx), 12, 6 + k);  // + contact twist cotangents
        }
    });
    ex.run([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        for (int i = lane; i < c.d.L; i += Exec::NL) {
            int type, ds;
            if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
                const DsimTopoRegs& tp = ex.topo(lane);
                type = tp.own_type; ds = tp.own_ds;
                DSIM_OPAQUE(type);
            } else {
                type = CI(jtype)[i]; ds = CI(qdstart)[i];
            }
            const sv6 a_vj = ldsv(WF(avj) + 6 * i) + ldsv(WF(avtot) + 6 * i);
            sv6 Z = ldsv(WF(aw) + 6 * i);
            if (c.d.NS > 0) Z += ldsv(WF(agx) + 12 * i);
            sv6 Wp = zerosv();
            float* aqd = WF(aqd);
            // operands of the hinge and free cases, fetched unconditionally with the loads above (one round trip)
            const sv6 S0 = ldsv(WF(S) + 6 * ds), aS0 = ldsv(WF(aS) + 6 * ds);
            const float qd0 = WF(qd)[ds], g0 = aqd[ds];
            // vj = S qd: cotangents of qd and of S; S is attached to the joint frame X_sj: W_par = sum S x* adj_S
            if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                const sv6 g6 = mksv(mk3(g0, aqd[ds + 1], aqd[ds + 2]), ld3(aqd + ds + 3));
                if (type == DSIM_JOINT_FREE) stsv(aqd + ds, g6 + a_vj);
            }
            if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
                Wp = scross_dual(S0, aS0 + a_vj * qd0);
                aqd[ds] = g0 + sdot(S0, a_vj);
            }
            if constexpr ((MASK & DSIM_TM(DSIM_JOINT_BALL)) != 0) {
                if (type == DSIM_JOINT_BALL) {
                    for (int k = 0; k < 3; ++k) {
                        const sv6 S = ldsv(WF(S) + 6 * (ds + k)), aS = ldsv(WF(aS) + 6 * (ds + k));
                        const float qd = WF(qd)[ds + k], g = aqd[ds + k];
                        Wp += scross_dual(S, aS + a_vj * qd);
                        aqd[ds + k] = g + sdot(S, a_vj);
                    }
                }
            }
            stsv(WF(aw) + 6 * i, Z + Wp);
            stsv(WF(awp) + 6 * i, Wp);
        }
    });
    ex.run([&](int lane) {
        if constexpr (DsimTrunk<Ctx, Exec>::value) {
            dsim_trunk_sum<true>(c, ex, lane, WF(aw), WF(acx), 12, 0, WF(azs));
            return;
        }
        for (int it = lane; it < 6 * c.d.L; it += Exec::NL) {
            const int i = it / 6, k = it - 6 * i;
            int n_known = -1;
            if constexpr (DsimSixRegs<Ctx, Exec::NL>::value) n_known = ex.topo(lane).six_n;
            (void)n_known;
            WF(azs)[it] = dsim_subtree_contact_sum(c, ex, lane, i, WF(aw), k, WF(acx), 12, k);  // + contact pose wrenches
        }
    });
    // adj q_d = S_d . (subtree wrench of the link, without the part attached to the link's own joint frame)
    ex.run([&](int lane) {
        constexpr int MASK = dsim_tmask_static<Ctx>();
        for (int i = lane; i < c.d.L; i += Exec::NL) {
            int type, cs, ds, par;
            if constexpr (DsimRoleRegs<Ctx, Exec::NL>::value) {
                const DsimTopoRegs& tp = ex.topo(lane);
                type = tp.own_type; cs = tp.own_cs; ds = tp.own_ds; par = tp.own_parent;
                DSIM_OPAQUE(type);
            } else {
                type = CI(jtype)[i]; cs = CI(qstart)[i]; ds = CI(qdstart)[i]; par = CI(parent)[i];
            }
            const sv6 Wt = ldsv(WF(azs) + 6 * i) - ldsv(WF(awp) + 6 * i);
            float* aq = WF(aq);
            const sv6 S0 = ldsv(WF(S) + 6 * ds);   // hinge operands, fetched unconditionally with the loads above
            const float g0 = aq[cs];
            if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) aq[cs] = g0 + sdot(S0, Wt);
            if constexpr ((MASK & DSIM_TM(DSIM_JOINT_BALL)) != 0) {
                if (type == DSIM_JOINT_BALL) {
                    const q4 r = ldq(WF(q) + cs), g = ldq(aq + cs);
                    const v3 t = mk3(sdot(ldsv(WF(S) + 6 * ds), Wt), sdot(ldsv(WF(S) + 6 * ds + 6), Wt),
                                     sdot(ldsv(WF(S) + 6 * ds + 12), Wt));
                    stq(aq + cs, g + qmul_v(t, r) * 2.0f);
                }
            }
            if constexpr ((MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0) {
                if (type == DSIM_JOINT_FREE) {
                    // X_sc = X_sj o (q_p, q_r): translation by R_j dq_p, rotation about p_c
                    const v3 pc = ld3(WF(xsc) + 7 * i);
                    const q4 rc = ldq(WF(xsc) + 7 * i + 3);
                    const v3 gp = mk3(g0, aq[cs + 1], aq[cs + 2]);
                    const q4 gr = ldq(aq + cs + 3);
                    const v3 tc = Wt.w - cross(pc, Wt.v);
                    if constexpr (DsimFreeRootIdent<Ctx>::value) {
                        // the only free joint is the root, with an identity joint frame: R_j = 1, and 1^-1 x = x, 1* (x) t = t exactly
                        (void)par;
                        st3(aq + cs, gp + Wt.v);
                        stq(aq + cs + 3, gr + qmul_v(tc, rc) * 2.0f);
                    } else {
                        const q4 rpj = ldq(CF(xpj) + 7 * i + 3);
                        q4 rj = rpj;
                        if (par >= 0) rj = qmul(ldq(WF(xsc) + 7 * par + 3), rpj);
                        st3(aq + cs, gp + rotate_inv(rj, Wt.v));
                        stq(aq + cs + 3, gr + qmul(qmul(qconj(rj), mkq(tc.x, tc.y, tc.z, 0.f)), rc) * 2.0f);
                    }
                }
            }
        }
    });
}

// Body level of the adjoint substep for row trees (DsimRowTree): ONE phase on the link lanes.  dsim_bwd_bodies below is the
// general form -- per-link block, subtree sum, per-link block, subtree sum, per-link block, subtree sum, per-link block: seven
// phases, each paying an LDS round trip for values that only change lanes inside the tree.  Here the three subtree sums are
// row-tree sums on registers and everything a link lane computes stays in its registers from the first load to the cotangents of
// its joint coordinates.  The one thing that comes from other lanes' ITEMS is the contact terms: the side block (the helper
// wavefront where there is one) evaluates contacts^T per contact and reduces them per body (agx: pose wrench 6, twist cotangent
// 6); the link lanes pick their body's row up at Exec::side_done(), after the first third of their work.
// What the wavefronts behind the first one do at the END of an adjoint substep's body level when the two run side by side
// (DsimWideOverlap): bring in the checkpoint row of the NEXT adjoint substep -- requested one substep earlier, Exec::prefetch_rest --
// and, where that substep starts a mass-matrix group, the inverse the group's forward substeps used; then request the row after it.
// Nothing the first wavefront still does in this substep reads the saved block or the inverse from LDS.
struct DsimNextRow {
    bool any = false;             // there is a next substep (its row waits in the prefetch registers)
    const float* hv = nullptr;    // ... and it enters a new group: that group's inverse (global memory)
    const float* after = nullptr; // the row to request afterwards (null: none left)
};
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_bodies_rowtree(const Ctx& c, Exec& ex, bool update_mass, const DsimNextRow& nx = DsimNextRow{}) {
    ex.mark(10);
    using D = decltype(c.d);
    constexpr int L = D::L, MASK = D::tmask;
    constexpr bool HAS_FREE = (MASK & DSIM_TM(DSIM_JOINT_FREE)) != 0, HAS_BALL = (MASK & DSIM_TM(DSIM_JOINT_BALL)) != 0;
    constexpr bool MUSCLES = D::NS > 0;   // per-item / per-body phases on all wavefronts, the body level on the first one
    constexpr bool WIDE = MUSCLES && DsimWideOverlap<Ctx, Exec>::value;   // ... at the same time
    auto fm = [&](int lane) __attribute__((always_inline)) {
        const DsimTopoRegs& tp = ex.topo(lane);
        const bool on = lane < L;
        const int i = on ? lane : 0;
        int type = tp.own_type;
        DSIM_OPAQUE(type);
        const int cs = tp.own_cs, ds = tp.own_ds, par = tp.own_parent;
        const bool hinge = type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE, fr = type == DSIM_JOINT_FREE;
        const bool ball = HAS_BALL && type == DSIM_JOINT_BALL;
        // ---- every operand that does not come out of this phase, in one round trip
        inertia10 I = ld_i10(WF(i10) + 10 * i);
        sv6 v = ldsv(WF(v) + 6 * i), a = ldsv(WF(a) + 6 * i), r = ldsv(WF(af) + 6 * i);
        const v3 grav = ld3(CF(grav));
        float g[10];
        for (int k = 0; k < 10; ++k) g[k] = update_mass ? WF(ai10m)[10 * i + k] : 0.f;
        sv6 S0 = ldsv(WF(S) + 6 * ds), aS0 = ldsv(WF(aS) + 6 * ds);
        float* aq = WF(aq);
        float* aqd = WF(aqd);
        const float qd0 = WF(qd)[ds], g0 = aqd[ds], gq0 = aq[cs];
        sv6 qd6 = zerosv(), g6 = zerosv();
        v3 pc = zero3(), gp = zero3();
        q4 rc = mkq(0.f, 0.f, 0.f, 1.f), gr = mkq(0.f, 0.f, 0.f, 0.f);
        if constexpr (HAS_FREE) {
            qd6 = mksv(mk3(qd0, WF(qd)[ds + 1], WF(qd)[ds + 2]), ld3(WF(qd) + ds + 3));
            g6 = mksv(mk3(g0, aqd[ds + 1], aqd[ds + 2]), ld3(aqd + ds + 3));
            pc = ld3(WF(xsc) + 7 * i);
            rc = ldq(WF(xsc) + 7 * i + 3);
            gp = mk3(gq0, aq[cs + 1], aq[cs + 2]);
            gr = ldq(aq + cs + 3);
        }
        // ball joints: the other two columns of the motion subspace, their cotangents, the joint's quaternion and its cotangent
        // (fetched by every lane: the words behind a hinge's single dof / coordinate are in range and unused)
        sv6 S1 = zerosv(), S2 = zerosv(), aS1 = zerosv(), aS2 = zerosv();
        float qd1 = 0.f, qd2 = 0.f, g1 = 0.f, g2 = 0.f;
        q4 rb = mkq(0.f, 0.f, 0.f, 1.f), gb = mkq(0.f, 0.f, 0.f, 0.f);
        if constexpr (HAS_BALL) {
            S1 = ldsv(WF(S) + 6 * ds + 6); S2 = ldsv(WF(S) + 6 * ds + 12);
            aS1 = ldsv(WF(aS) + 6 * ds + 6); aS2 = ldsv(WF(aS) + 6 * ds + 12);
            qd1 = WF(qd)[ds + 1]; qd2 = WF(qd)[ds + 2];
            g1 = aqd[ds + 1]; g2 = aqd[ds + 2];
            rb = ldq(WF(q) + cs);
            gb = mkq(gq0, aq[cs + 1], aq[cs + 2], aq[cs + 3]);
        }
        // the pose of the parent link's frame (free joints in a rotated joint frame only; read before the side block is waited for)
        q4 rpar = mkq(0.f, 0.f, 0.f, 1.f);
        if constexpr (HAS_FREE && !DsimFreeRootIdent<Ctx>::value) rpar = ldq(WF(xsc) + 7 * (par >= 0 ? par : 0) + 3);
        ex.loads_landed();
        // ---- f^T, velocity / acceleration recursions^T of the link itself, pose wrench of inertia + gravity (dsim_bwd_bodies)
        const sv6 hv = inertia_mul(I, v);
        sv6 a_v, a_hv;
        a_v.w = cross(hv.w, r.w) + cross(hv.v, r.v);
        a_v.v = cross(hv.v, r.w);
        a_hv.w = cross(r.w, v.w);
        a_hv.v = cross(r.w, v.v) + cross(r.v, v.w);
        a_v += inertia_mul(I, a_hv);
#ifdef DSIM_INJECT_ADJ_ERROR
        a_v = a_v * DSIM_INJECT_ADJ_ERROR;   // (developer / test builds only, see dsim_bwd_bodies)
#endif
        inertia_bilinear_adj(g, r, a, 1.0f);
        inertia_bilinear_adj(g, a_hv, v, 1.0f);
        sv6 W = inertia_pose_wrench(I, g);
        const v3 rg = cross(r.w, grav);
        W.w += cross(I.h, rg);
        W.v += rg * I.m;
        sv6 A = inertia_mul(I, r);   // aa
        // Lanes without a link computed on link 0's operands: finite values that no sum picks up (a step's weight is 1 only where
        // lane + distance is a child, i.e. a link).  They are zeroed all the same: measured (tools/ab_min.py, Ant 1024, same box,
        // alternating runs) the adjoint launch takes 0.0538 ms with these 18 selects and 0.0586 ms without them -- 25 instructions
        // more and 8 % faster; zeroing the lanes' OPERANDS instead changes nothing (0.0582 ms), so it is what the selects do to
        // the compiler's schedule of the block, not the data.  Kept as measured.
        if (!on) {
            A = zerosv();
            a_v = zerosv();
            W = zerosv();
        }
        // ---- aatot = subtree sum of aa; joint motion^T
        // (several wavefronts: still in front of the first hand-over -- the other wavefronts' per-item block is the longer side there,
        // measured by leaving it out: 11 % of the adjoint launch)
        dsim_rowtree_sum(c, ex, tp, A);
        sv6 vj = zerosv();
        if constexpr (HAS_FREE) {
            if (fr) vj = qd6;
        }
        if (hinge) vj = S0 * qd0;
        if constexpr (HAS_BALL) {
            if (ball) {
                vj = S0 * qd0;
                vj += S1 * qd1;
                vj += S2 * qd2;
            }
        }
        a_v.w += cross(vj.w, A.w) + cross(vj.v, A.v);
        a_v.v += cross(vj.w, A.v);
        sv6 a_vj;
        a_vj.w = cross(A.w, v.w) + cross(A.v, v.v);
        a_vj.v = cross(A.v, v.w);
        ex.stamp();
        // ---- the contact (and muscle) terms of the own body, reduced per body by the side block / the phases before this one /
        // the other wavefronts
        // (several wavefronts: two hand-overs -- the contacts' twist cotangents per body need no muscle chunk sums and come first,
        // mid2; the pose wrenches per body, side_done_w, are picked up further down, in front of the last subtree sum)
        if constexpr (WIDE) {
            ex.mid();    // (the other wavefronts: per-item cotangents done, the contacts' twist cotangents are gathered per body ...
            ex.mid2();   // ... and are there)
        } else {
            ex.side_done();
        }
        sv6 cpose = zerosv();
        if constexpr (!WIDE) cpose = ldsv(WF(agx) + 12 * i);
        const sv6 ctw = ldsv(WF(agx) + 12 * i + 6);
        ex.loads_landed();
        // ---- avtot = subtree sum of (a_v + contact twist cotangents); cotangents of qd and of S; S is attached to the joint frame
        sv6 T = a_v + ctw;
        dsim_rowtree_sum(c, ex, tp, T);
        a_vj += T;
        if constexpr (WIDE) ex.mid2();   // (the other wavefronts: chunk sums of the muscle rows done -- beside the subtree sum above --, the per-body pose gather may start)
        sv6 Wp = zerosv();
        float aqd_new = g0, aqd_new1 = g1, aqd_new2 = g2;
        if (hinge) {
            Wp = scross_dual(S0, aS0 + a_vj * qd0);
            aqd_new = g0 + sdot(S0, a_vj);
        }
        if constexpr (HAS_BALL) {
            if (ball) {
                Wp = scross_dual(S0, aS0 + a_vj * qd0);
                Wp += scross_dual(S1, aS1 + a_vj * qd1);
                Wp += scross_dual(S2, aS2 + a_vj * qd2);
                aqd_new = g0 + sdot(S0, a_vj);
                aqd_new1 = g1 + sdot(S1, a_vj);
                aqd_new2 = g2 + sdot(S2, a_vj);
            }
        }
        // ---- azs = subtree sum of the pose wrenches (inertia + gravity, joint frame of the children's S, contacts / muscles)
        if constexpr (WIDE) {
            ex.side_done_w();   // the per-body pose wrenches of muscles + contacts are there
            cpose = ldsv(WF(agx) + 12 * i);
            ex.loads_landed();
        }
        sv6 Z = W + Wp + cpose;
        dsim_rowtree_sum(c, ex, tp, Z);
        const sv6 Wt = Z - Wp;   // without the part attached to the link's own joint frame
        ex.stamp();
        // ---- adj q_d = S_d . Wt; all stores of the phase
        if (on) {
            if (hinge) {
                aqd[ds] = aqd_new;
                aq[cs] = gq0 + sdot(S0, Wt);
            }
            if constexpr (HAS_BALL) {
                if (ball) {
                    aqd[ds] = aqd_new;
                    aqd[ds + 1] = aqd_new1;
                    aqd[ds + 2] = aqd_new2;
                    const v3 t = mk3(sdot(S0, Wt), sdot(S1, Wt), sdot(S2, Wt));
                    stq(aq + cs, gb + qmul_v(t, rb) * 2.0f);
                }
            }
            if constexpr (HAS_FREE) {
                if (fr) {
                    stsv(aqd + ds, g6 + a_vj);
                    const v3 tc = Wt.w - cross(pc, Wt.v);
                    if constexpr (DsimFreeRootIdent<Ctx>::value) {
                        (void)par;
                        st3(aq + cs, gp + Wt.v);
                        stq(aq + cs + 3, gr + qmul_v(tc, rc) * 2.0f);
                    } else {
                        const q4 rpj = ldq(CF(xpj) + 7 * i + 3);
                        q4 rj = rpj;
                        if (par >= 0) rj = qmul(rpar, rpj);
                        st3(aq + cs, gp + rotate_inv(rj, Wt.v));
                        stq(aq + cs + 3, gr + qmul(qmul(qconj(rj), mkq(tc.x, tc.y, tc.z, 0.f)), rc) * 2.0f);
                    }
                }
            }
        }
    };
    if constexpr (WIDE) {
        // first wavefront: the body level; the others, at the same time: per item contacts^T, muscles^T | per muscle the activation
        // cotangent, chunk sums of the muscle rows | per body: muscle + contact pose wrenches (6), contact twist cotangents (6) |
        // the next substep's checkpoint row
        constexpr int NLR = Exec::NL - DSIM_NL;
        ex.fork_wave0([&](int lane) {
            fm(lane);
            if (nx.hv) dsim_hacc_zero(c, ex, lane);
        }, [&](int lane) {
            dsim_bwd_external_items_n<NLR>(c, ex, lane);
            ex.mid();
            for (int it = lane; it < 6 * c.d.L; it += NLR) {   // per body: twist cotangents of its contacts (no muscle terms)
                const int i = it / 6, r = 6 + it - 6 * i;
                WF(agx)[12 * i + r] = dsim_body_contact_sum(c, i, WF(acx), 12, r, 0.f);
            }
            ex.mid2();
            // (the chunk sums are needed by the pose gather only: behind the hand-over of the twist cotangents, which the first
            // wavefront waits for -- leaving them in front of it cost 6 % of the launch)
            dsim_muscle_chunk_sums<8>(c, lane, NLR);
            ex.mid2();
            for (int it = lane; it < 6 * c.d.L; it += NLR) {   // per body: pose wrenches of its muscle rows (chunk sums) + its contacts
                const int i = it / 6, r = it - 6 * i;
                WF(agx)[12 * i + r] = dsim_body_contact_sum(c, i, WF(acx), 12, r, dsim_body_chunk_sum(c, i, r, 0.f));
            }
            ex.side_done_w();
            // nothing the first wavefront still needs: the activation cotangents (one sum per muscle) and the next row
            for (int m = lane; m < c.d.M; m += NLR) {
                const float gm = WF(amact)[m];
                WF(amact)[m] = gm + dsim_muscle_act_sum(c, m);
            }
            if (nx.any) {
                ex.commit_rest(WF(q), dsim_row(c), lane);
                if (nx.hv) {
                    for (int k = lane; k < c.d.nd * c.d.nd; k += NLR) {
                        WF(hinv)[k] = nx.hv[k];
                        WF(aH)[k] = 0.f;
                    }
                    dsim_hacc_zero(c, ex, lane + DSIM_NL);
                }
                if (nx.after) ex.prefetch_rest(nx.after, dsim_row(c));
            }
        });
    } else if constexpr (MUSCLES) {
        // per item (all wavefronts): contacts^T, muscles^T; per muscle: the activation cotangent; chunk sums of the muscle rows;
        // per body: muscle pose wrenches + contact pose wrenches (6), contact twist cotangents (6) -- the phases of dsim_bwd_bodies
        ex.run([&](int lane) { dsim_bwd_external_items(c, ex, lane); });
        ex.run([&](int lane) {
            for (int m = lane; m < c.d.M; m += Exec::NL) {
                const float gm = WF(amact)[m];
                WF(amact)[m] = gm + dsim_muscle_act_sum(c, m);
            }
            dsim_muscle_chunk_sums(c, lane, Exec::NL);
        });
        ex.run([&](int lane) {
            for (int it = lane; it < 12 * c.d.L; it += Exec::NL) {
                const int i = it / 12, r = it - 12 * i;
                float acc = 0.f;
                if (r < 6) acc = dsim_body_chunk_sum(c, i, r, 0.f);
                WF(agx)[it] = dsim_body_contact_sum(c, i, WF(acx), 12, r, acc);
            }
        });
        ex.run_wave0(fm);
    } else {
        ex.fork_side(fm, [&](int lane) {
            // side block: contacts^T per contact, then the per-body sums of their 12-float rows (bodies without contacts: zeros)
            dsim_bwd_external_items(c, ex, lane);
            ex.lds_fence();
            const DsimTopoRegs& tp = ex.topo(lane);
            constexpr int GXP = (12 * L + Exec::NL - 1) / Exec::NL, CB = D::CBMAX > 0 ? D::CBMAX : 1;
            float x[GXP][CB];
#pragma unroll
            for (int p = 0; p < GXP; ++p)   // all loads first: addresses are register + immediate (rows past a body's last contact are
#pragma unroll                              // other contacts' rows or the arrays behind acx: selected away)
                for (int e = 0; e < CB; ++e) x[p][e] = WF(acx)[tp.gx_row[p] + 12 * e];
#pragma unroll
            for (int p = 0; p < GXP; ++p) {
                float acc = 0.f;
                int n = tp.gx_n[p];
                DSIM_OPAQUE(n);   // (the e < n masks are recomputed here, not kept as loop invariants in spilled SGPR pairs)
#pragma unroll
                for (int e = 0; e < CB; ++e) acc += (e < n) ? x[p][e] : 0.f;
                const int it = lane + Exec::NL * p;
                if (it < 12 * L) WF(agx)[it] = acc;
            }
        });
    }
}

// Lean checkpoint mode: the row holds only (q, qd); the forward intermediates of the substep are recomputed here with the
// forward pass's own phases (same code, same inputs: bit-identical to what the full mode reads back from HBM).
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_recompute_forward(const Ctx& c, Exec& ex) {
    dsim_fwd_kinematics(c, ex);
    dsim_fwd_external(c, ex);
    dsim_fwd_tau(c, ex);
    dsim_fwd_solve(c, ex);
}

// nx: DsimWideOverlap only (the body level brings the next substep's checkpoint row in, see DsimNextRow)
// g_lit_aH (dsim_step_backward_literal only; null everywhere else, folded away): where the refresh substep leaves the accumulated
// cotangent of the mass matrix [nd][nd] before the mass-matrix adjoint consumes it (dsim_literal.hpp)
template <class Ctx, class Exec> DSIM_FN void dsim_bwd_substep(const Ctx& c, Exec& ex, bool update_mass, const DsimNextRow& nx = DsimNextRow{},
                                                               float* g_lit_aH = nullptr) {
    dsim_bwd_joint_space(c, ex, update_mass);
    if (update_mass && g_lit_aH)
        ex.fire([&](int lane) {
            for (int k = lane; k < c.d.nd * c.d.nd; k += Exec::NL) g_lit_aH[k] = WF(aH)[k];
        });
    if (update_mass) dsim_bwd_mass(c, ex);
    if constexpr (DsimRowTree<Ctx, Exec>::value) dsim_bwd_bodies_rowtree(c, ex, update_mass, nx);
    else dsim_bwd_bodies(c, ex, update_mass);
}
// the body level of the adjoint substeps commits the checkpoint rows itself (first wavefront: body level, the others: items and rows)
template <class Ctx, class Exec> struct DsimWideRows {
    static constexpr bool value = []() {
        if constexpr (DsimRowTree<Ctx, Exec>::value) return DsimWideOverlap<Ctx, Exec>::value;
        else return false;
    }();
};

// Reverse sweep of one env.step().  g_ckpt is this environment's [substeps][nq+nd] checkpoint.
template <class Ctx, class Exec>
DSIM_FN void dsim_sim_step_backward(const Ctx& c, Exec& ex, int substeps, int mm_freq, const float* g_ckpt,
                                    const float* g_act, const float* g_mact, const float* g_gq_out,
                                    const float* g_gqd_out, float* g_gq_in, float* g_gqd_in, float* g_gact,
                                    float* g_gmact, float* g_lit = nullptr) {
    // g_lit (dsim_step_backward_literal): this environment's [nq + nd + nd * nd] words -- the cotangents of the FIRST substep's
    // outputs (q_1, qd_1) and the mass-matrix cotangent of the first group, what dsim_literal.hpp contracts its tangent with
    const int nq = c.d.nq, nd = c.d.nd, M = c.d.M;
    ex.begin_request();
    ex.begin();
    dsim_init_static(c, ex);
    ex.run_both([&](int lane) { dsim_topo_init(c, ex, lane); });   // every wave keeps its own topology records
    ex.run([&](int lane) {
        for (int k = lane; k < nq; k += Exec::NL) WF(aqn)[k] = g_gq_out[k];
        for (int k = lane; k < nd; k += Exec::NL) {
            WF(aqdn)[k] = g_gqd_out[k];
            WF(act)[k] = g_act[k];
            WF(aact)[k] = 0.f;
        }
        for (int k = lane; k < M; k += Exec::NL) {
            WF(mact)[k] = g_mact[k];
            WF(amact)[k] = 0.f;
        }
    });
    const int groups = (substeps + mm_freq - 1) / mm_freq;
    for (int g = groups - 1; g >= 0; --g) {
        const int s0 = g * mm_freq, s1 = (s0 + mm_freq < substeps) ? s0 + mm_freq : substeps;
        for (int s = s1 - 1; s >= s0; --s) {
            // forward intermediates of substep s (and, entering a group, the inverse its substeps used) from HBM.
            // The row was requested one substep earlier (ex.prefetch keeps it in flight in registers while the
            // previous adjoint substep computes), so this phase only moves registers to LDS.
            const float* hv = (s == s1 - 1) ? dsim_ckpt_hinv(c, const_cast<float*>(g_ckpt), substeps, g) : nullptr;
            if constexpr (DsimHelperCommit<Ctx, Exec>::value) {
                if (s == substeps - 1) {
                    ex.helper_prefetch(g_ckpt + (size_t)s * dsim_row(c), dsim_row(c));
                    ex.group_sync();   // the main wave is done with the prologue's use of q / qd (the words the row lands on)
                }
                ex.helper_commit(WF(q), dsim_row(c), s > 0 ? g_ckpt + (size_t)(s - 1) * dsim_row(c) : nullptr);
                if (hv)
                    ex.run([&](int lane) {
                        for (int k = lane; k < nd * nd; k += Exec::NL) {
                            WF(hinv)[k] = hv[k];
                            WF(aH)[k] = 0.f;
                        }
                        dsim_hacc_zero(c, ex, lane);
                    });
            } else if constexpr (DsimWideRows<Ctx, Exec>::value) {
                // only the launch's first row is committed here; every later one by the body level of the substep before it
                if (s == substeps - 1) {
                    ex.prefetch(g_ckpt + (size_t)s * dsim_row(c), dsim_row(c));
                    ex.run([&](int lane) {
                        ex.commit(WF(q), dsim_row(c), lane);
                        for (int k = lane; k < nd * nd; k += Exec::NL) {
                            WF(hinv)[k] = hv[k];
                            WF(aH)[k] = 0.f;
                        }
                        dsim_hacc_zero(c, ex, lane);
                    });
                    if (s > 0) ex.prefetch_rest(g_ckpt + (size_t)(s - 1) * dsim_row(c), dsim_row(c));
                }
            } else {
                if (s == substeps - 1) ex.prefetch(g_ckpt + (size_t)s * dsim_row(c), dsim_row(c));
                ex.run([&](int lane) {
                    ex.commit(WF(q), dsim_row(c), lane);
                    if (hv) {
                        for (int k = lane; k < nd * nd; k += Exec::NL) {
                            WF(hinv)[k] = hv[k];
                            WF(aH)[k] = 0.f;
                        }
                        dsim_hacc_zero(c, ex, lane);
                    }
                });
                if (s > 0) ex.prefetch(g_ckpt + (size_t)(s - 1) * dsim_row(c), dsim_row(c));
            }
            if constexpr (Ctx::LEAN) dsim_bwd_recompute_forward(c, ex);
            if (s == s0) dsim_fwd_composite(c, ex);
            DsimNextRow nx;
            if constexpr (DsimWideRows<Ctx, Exec>::value) {
                nx.any = s > 0;
                nx.hv = (s == s0 && g > 0) ? dsim_ckpt_hinv(c, const_cast<float*>(g_ckpt), substeps, g - 1) : nullptr;
                nx.after = s > 1 ? g_ckpt + (size_t)(s - 2) * dsim_row(c) : nullptr;
            }
            if (g_lit && s == 0)
                ex.fire([&](int lane) {
                    for (int k = lane; k < nq; k += Exec::NL) g_lit[k] = WF(aqn)[k];
                    for (int k = lane; k < nd; k += Exec::NL) g_lit[nq + k] = WF(aqdn)[k];
                });
            dsim_bwd_substep(c, ex, s == s0, nx, (g_lit && s == 0) ? g_lit + nq + nd : nullptr);  // aq / aqd (== aqn / aqdn, same LDS words) now belong to substep s - 1
        }
    }
    ex.run([&](int lane) {
        for (int k = lane; k < nq; k += Exec::NL) g_gq_in[k] = WF(aqn)[k];
        for (int k = lane; k < nd; k += Exec::NL) {
            g_gqd_in[k] = WF(aqdn)[k];
            if (g_gact) g_gact[k] = WF(aact)[k];
        }
        if (g_gmact)
            for (int k = lane; k < M; k += Exec::NL) g_gmact[k] = WF(amact)[k];
    });
}

// ================================================================================================
// fused environment surface: action clip/scale -> joint_act / muscle
// activations, observation vector and reward computed by the step kernels, adjoint fused likewise.
// Reference: envs/ant.py:156-174,266-307, envs/humanoid.py:186-215,314-356,
// envs/snu_humanoid.py:243-275,376-416, envs/cartpole_swing_up.py:113-125,204-225.
// ================================================================================================
#define DSIM_ENV_LOCOMOTION 1   // free-floating root: [h, quat, lin vel, ang vel, joint q, joint qd*s, up, heading, (actions)]
#define DSIM_ENV_CARTPOLE 2     // [x, xdot, sin th, cos th, thdot]
#define DSIM_ENV_PLANAR 3       // planar root (slide x, slide z, hinge y): [q[1:], qd]  (hopper, half-cheetah)
#define DSIM_REW_ANT 0
#define DSIM_REW_HUMANOID 1
#define DSIM_REW_SNU 2
#define DSIM_REW_CARTPOLE 3
#define DSIM_REW_HOPPER 4
#define DSIM_REW_CHEETAH 5

struct DsimEnvSpec {
    int kind, rew_kind;
    int n_act, n_obs;
    int act_offset;      // joint_act[act_offset + k] = clip(a_k) * act_scale[k]            (act_muscle == 0)
    int act_muscle;      // muscle_act[k] = (clip(a_k) * 0.5 + 0.5) * act_scale[k]           (act_muscle == 1)
    int obs_actions;     // append the (clipped / remapped) actions to the observation
    int sanitize;        // the adjoint launch returns 0 for non-finite cotangents (include/dsim.h: sanitize_grads)
    float isr[4];        // conjugate of the start rotation
    float tgt_x, tgt_z;  // targets + start_pos (x, z)
    float term_h, term_tol, h_scale, act_pen, vel_scale;
    float pen[4];        // cartpole: pole angle, pole velocity, cart position, cart velocity penalties
    const float* act_scale;  // [n_act], device memory
};

// Episode bookkeeping fused into the step.  The reference does this with torch ops and a device->host sync per step:
// progress_buf / reset_buf at the end of calculateReward (envs/ant.py:297-307, humanoid.py:340-356, hopper.py:288-293)
// and reset(env_ids) (ant.py:192-234).  Here the step kernel flags the finished environments itself and restarts
// them from a pool of start states the host drew from the environment's own reset distribution; entry
// (reset_count[e] % pool) is consumed.  progress == nullptr: no episode handling (plain step).
struct DsimEpisode {
    long long* progress;    // [N] in/out: progress_buf
    long long* done;        // [N] out: reset_buf
    float* obs_before;      // [N][n_obs] out or nullptr: observation before the reset (extras['obs_before_reset'])
    const float* reset_q;   // [pool][N][nq]
    const float* reset_qd;  // [pool][N][nd]
    int* reset_count;       // [N] in/out
    int pool, episode_length, height_terminate, check_invalid;
    const float* noise_q;   // [nq] or nullptr: in-kernel stochastic restart (include/dsim.h: dsim_episode)
    const float* noise_qd;  // [nd] or nullptr
    float noise_angle;
    unsigned long long seed;
};

// Philox4x32-10 (Salmon et al., SC'11): counter-based generator, the one torch.rand uses on GPUs.  Restart noise of
// environment e, restart number n, coordinate stream w: counter (e, n, w, 0), key = seed; 4 uniforms in [0, 1) with 24 bits.
DSIM_FN void dsim_philox4(unsigned long long seed, unsigned e, unsigned n, unsigned w, float* u) {
    unsigned c0 = e, c1 = n, c2 = w, c3 = 0u, k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    u[0] = (float)(c0 >> 8) * (1.0f / 16777216.0f);
    u[1] = (float)(c1 >> 8) * (1.0f / 16777216.0f);
    u[2] = (float)(c2 >> 8) * (1.0f / 16777216.0f);
    u[3] = (float)(c3 >> 8) * (1.0f / 16777216.0f);
}
#define DSIM_EP_DONE 1.0f
#define DSIM_EP_INVALID 3.0f  // finished because the state blew up: reward forced to 0 (humanoid.py:340-356)

// ---- early loads -------------------------------------------------------------------------------------------------------
// The prologue of a launch reads a handful of small rows from global memory (state, actions, action scales, cotangents).
// Read where they are used, every one of them is a full memory latency on the critical path of a launch that lasts only
// ~30 of them (measured: 17.5 k of the Ant adjoint's 245 k cycles, 5.5 k of the forward's 131 k).  The specialised kernels
// therefore REQUEST them all first, into per-lane registers (Exec::io), then let the model constants arrive (Exec::begin),
// and the prologue phases find their inputs in registers: one latency for everything.  Row lengths are compile-time
// bounds there; the generic kernels (run-time sizes) read at the point of use as before.
#define DSIM_IO_MAX 24
template <class Ctx, class Exec> struct DsimIo {
    static constexpr bool PRE = DsimIsStatic<Ctx>::value;
    static constexpr int NL = Exec::NL;
    static constexpr int dim(int which) {
        if constexpr (DsimIsStatic<Ctx>::value) {
            using D = decltype(Ctx::d);
            return which == 0 ? D::nq : (which == 1 ? D::nd : (D::M > D::nd ? D::M : D::nd));
        } else {
            return 0;
        }
    }
    static constexpr int cdiv(int a) { return (a + NL - 1) / NL; }
    static constexpr int CQ = cdiv(dim(0)), CD = cdiv(dim(1)), CA = cdiv(dim(2)), CO = cdiv(16 + dim(0) + dim(1) + dim(2));
    // register map (floats per lane)
    static constexpr int Q = 0, QD = Q + CQ, GQ = QD + CD, GQD = GQ + CQ, ACT = GQD + CD, SCALE = ACT + CA, GOBS = SCALE + CA,
                         GOBSB = GOBS + CO, GREW = GOBSB + CO, END = GREW + 1;
    static_assert(!PRE || END <= DSIM_IO_MAX, "early-load registers: raise DSIM_IO_MAX");
};
// f(k, u) for this lane's entries k = lane + NL u < n of a row.  CAP > 0: compile-time trip count, so that u is a constant
// after unrolling and io[... + u] a register (a run-time index would put the whole array into scratch memory);
// CAP == 0 (generic kernels): the run-time loop, u unused.
template <int CAP, int NL, class F> DSIM_FN void dsim_io_each(int n, int lane, F&& f) {
    if constexpr (CAP > 0) {
#pragma unroll
        for (int u = 0; u < CAP; ++u) {
            const int k = lane + NL * u;
            if (k < n) f(k, u);
        }
    } else {
        for (int k = lane; k < n; k += NL) f(k, 0);
    }
}
// r[u] = g[lane + NL u] (0 where the row ends or g is null)
template <int CAP, int NL> DSIM_FN void dsim_io_fetch(float* r, const float* g, int n, int lane) {
#pragma unroll
    for (int u = 0; u < CAP; ++u) {
        const int k = lane + NL * u;
        r[u] = (g && k < n) ? g[k] : 0.f;
    }
}

// actions -> LDS: ua (what the env stores as self.actions), act / mact
template <class Ctx, class Exec>
DSIM_FN void dsim_env_load_actions(const Ctx& c, Exec& ex, const DsimEnvSpec& sp, const float* g_actions, bool early = false) {
    using IO = DsimIo<Ctx, Exec>;
    ex.run([&](int lane) {
        for (int k = lane; k < c.d.nd; k += Exec::NL) WF(act)[k] = 0.f;
    });
    ex.run([&](int lane) {
        const float* io = ex.io(lane);
        auto one = [&](int k, float a, float sc) {
            a = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
            if (sp.act_muscle) {
                a = a * 0.5f + 0.5f;
                WF(mact)[k] = a * sc;
            } else {
                WF(act)[sp.act_offset + k] = a * sc;
            }
            WF(ua)[k] = a;
        };
        if (IO::PRE && early) dsim_io_each<IO::CA, Exec::NL>(sp.n_act, lane, [&](int k, int u) { one(k, io[IO::ACT + u], io[IO::SCALE + u]); });
        else dsim_io_each<0, Exec::NL>(sp.n_act, lane, [&](int k, int) { one(k, g_actions[k], sp.act_scale[k]); });
    });
}

// observation of the state in LDS (q, qd, ua) -> LDS obs
template <class Ctx, class Exec> DSIM_FN void dsim_env_obs_compute(const Ctx& c, Exec& ex, const DsimEnvSpec& sp) {
    const int nq = c.d.nq, nd = c.d.nd;
    ex.run([&](int lane) {
        const float *q = WF(q), *qd = WF(qd);
        float* o = WF(obs);
        if (sp.kind == DSIM_ENV_LOCOMOTION) {
            const int nj = nq - 7, njd = nd - 6;
            if (lane == 0) {
                const v3 pos = ld3(q), w = ld3(qd), vl = ld3(qd + 3);
                const q4 r = ldq(q + 3);
                o[0] = pos.y;
                stq(o + 1, r);
                st3(o + 5, vl - cross(pos, w));
                st3(o + 8, w);
                const q4 tq = qmul(r, mkq(sp.isr[0], sp.isr[1], sp.isr[2], sp.isr[3]));
                const v3 up = rotate(tq, mk3(0.f, 1.f, 0.f)), hd = rotate(tq, mk3(1.f, 0.f, 0.f));
                const float tx = sp.tgt_x - pos.x, tz = sp.tgt_z - pos.z;
                float l = sqrtf(tx * tx + tz * tz);
                l = l < 1e-9f ? 1e-9f : l;
                o[11 + nj + njd] = up.y;
                o[12 + nj + njd] = hd.x * (tx / l) + hd.z * (tz / l);
            }
            for (int k = lane; k < nj; k += Exec::NL) o[11 + k] = q[7 + k];
            for (int k = lane; k < njd; k += Exec::NL) o[11 + nj + k] = sp.vel_scale * qd[6 + k];
            if (sp.obs_actions)
                for (int k = lane; k < sp.n_act; k += Exec::NL) o[13 + nj + njd + k] = WF(ua)[k];
        } else if (sp.kind == DSIM_ENV_CARTPOLE) {
            if (lane == 0) {
                o[0] = q[0];
                o[1] = qd[0];
                o[2] = sinf(q[1]);
                o[3] = cosf(q[1]);
                o[4] = qd[1];
            }
        } else if (sp.kind == DSIM_ENV_PLANAR) {
            for (int k = lane; k < nq - 1; k += Exec::NL) o[k] = q[1 + k];
            for (int k = lane; k < nd; k += Exec::NL) o[nq - 1 + k] = qd[k];
        }
    });
}

// reward of the observation in LDS (one lane calls this)
template <class Ctx> DSIM_FN float dsim_env_reward(const Ctx& c, const DsimEnvSpec& sp) {
    const int nq = c.d.nq, nd = c.d.nd;
    const float* o = WF(obs);
    float r = 0.f;
    if (sp.kind == DSIM_ENV_LOCOMOTION) {
        const int iu = 11 + (nq - 7) + (nd - 6);
        r = o[5] + 0.1f * o[iu] + o[iu + 1];
        float pen = 0.f;
        if (sp.rew_kind == DSIM_REW_SNU) {
            for (int k = 0; k < sp.n_act; ++k) pen += fabsf(WF(ua)[k]);
        } else {
            for (int k = 0; k < sp.n_act; ++k) pen += WF(ua)[k] * WF(ua)[k];
        }
        r += pen * sp.act_pen;
        if (sp.rew_kind == DSIM_REW_ANT) {
            r += o[0] - sp.term_h;
        } else if (sp.rew_kind == DSIM_REW_HUMANOID) {
            float hr = o[0] - (sp.term_h + sp.term_tol);
            hr = hr < -1.0f ? -1.0f : (hr > sp.term_tol ? sp.term_tol : hr);
            if (hr < 0.0f) hr = -200.0f * hr * hr;
            if (hr > 0.0f) hr = sp.h_scale * hr;
            r += hr;
        }
    } else if (sp.kind == DSIM_ENV_CARTPOLE) {
        const float th = atan2f(o[2], o[3]);
        const float a = WF(ua)[0];
        r = -th * th * sp.pen[0] - o[4] * o[4] * sp.pen[1] - o[0] * o[0] * sp.pen[2] - o[1] * o[1] * sp.pen[3] -
            a * a * sp.act_pen;
    } else if (sp.kind == DSIM_ENV_PLANAR) {
        float pen = 0.f;
        for (int k = 0; k < sp.n_act; ++k) pen += WF(ua)[k] * WF(ua)[k];
        r = o[nq - 1] + pen * sp.act_pen;  // progress = qd[0]
        if (sp.rew_kind == DSIM_REW_HOPPER) {
            // envs/hopper.py:279-288: clipped / shaped height term + upright term (pen[0] = termination angle)
            float hr = o[0] - (sp.term_h + sp.term_tol);
            hr = hr < -1.0f ? -1.0f : (hr > 0.3f ? 0.3f : hr);
            if (hr < 0.0f) hr = -200.0f * hr * hr;
            if (hr > 0.0f) hr = sp.h_scale * hr;
            r += hr + (1.0f - o[1] * o[1] / (sp.pen[0] * sp.pen[0]));
        }
    }
    return r;
}

template <class Ctx, class Exec>
DSIM_FN void dsim_env_observe(const Ctx& c, Exec& ex, const DsimEnvSpec& sp, float* g_obs, float* g_rew) {
    dsim_env_obs_compute(c, ex, sp);
    ex.run([&](int lane) {
        const float* o = WF(obs);
        for (int k = lane; k < sp.n_obs; k += Exec::NL) g_obs[k] = o[k];
        if (lane == 0) g_rew[0] = dsim_env_reward(c, sp);
    });
}


// obs/reward^T: adds into aqn/aqdn (cotangents of the step's output state) and writes gua (d/d stored action)
template <class Ctx, class Exec>
DSIM_FN void dsim_env_observe_adjoint(const Ctx& c, Exec& ex, const DsimEnvSpec& sp, const float* g_gobs,
                                      const float* g_grew, const float* g_gobs_before, float ep_flags, bool early = false) {
    using IO = DsimIo<Ctx, Exec>;
    const int nq = c.d.nq, nd = c.d.nd;
    // q/qd in LDS hold the step's end state (before any reset), ua the stored actions.  Any cotangent pointer may be
    // null (= zeros).  An environment that was restarted by this step returned the observation of its NEW state as obs
    // (no dependence on this step) and the old one as obs_before_reset; an invalid state had its reward overwritten.
    const bool live = ep_flags == 0.f;
    ex.run([&](int lane) {
        const float* io = ex.io(lane);
        // obs buffer reused for its cotangent
        if (IO::PRE && early)
            dsim_io_each<IO::CO, Exec::NL>(sp.n_obs, lane, [&](int k, int u) {
                WF(obs)[k] = (live ? io[IO::GOBS + u] : 0.f) + io[IO::GOBSB + u];   // 0 where there is no such cotangent
            });
        else
            dsim_io_each<0, Exec::NL>(sp.n_obs, lane, [&](int k, int) {
                float g = (live && g_gobs) ? g_gobs[k] : 0.f;
                if (g_gobs_before) g += g_gobs_before[k];
                WF(obs)[k] = g;
            });
    });
    ex.run([&](int lane) {
        const float *q = WF(q), *qd = WF(qd);
        float gr_in;
        if (IO::PRE && early) gr_in = ex.io(lane)[IO::GREW];
        else gr_in = g_grew ? g_grew[0] : 0.f;
        const float gr = ep_flags != DSIM_EP_INVALID ? gr_in : 0.f;
        float* go = WF(obs);
        if (sp.kind == DSIM_ENV_LOCOMOTION) {
            const int nj = nq - 7, njd = nd - 6, iu = 11 + nj + njd;
            for (int k = lane; k < nj; k += Exec::NL) WF(aqn)[7 + k] += go[11 + k];
            for (int k = lane; k < njd; k += Exec::NL) WF(aqdn)[6 + k] += sp.vel_scale * go[11 + nj + k];
            for (int k = lane; k < sp.n_act; k += Exec::NL) {
                const float a = WF(ua)[k];
                float g = sp.obs_actions ? go[iu + 2 + k] : 0.f;
                if (sp.rew_kind == DSIM_REW_SNU) g += gr * sp.act_pen * (a < 0.0f ? -1.0f : (a > 0.0f ? 1.0f : 0.0f));
                else g += gr * sp.act_pen * 2.0f * a;
                WF(gua)[k] = g;
            }
            if (lane == 0) {
                const v3 pos = ld3(q), w = ld3(qd);
                const q4 r = ldq(q + 3);
                // reward -> obs cotangents
                float g_h = go[0], g_up = go[iu] + 0.1f * gr, g_hd = go[iu + 1] + gr;
                v3 g_lv = ld3(go + 5);
                g_lv.x += gr;
                if (sp.rew_kind == DSIM_REW_ANT) {
                    g_h += gr;
                } else if (sp.rew_kind == DSIM_REW_HUMANOID) {
                    const float d = pos.y - (sp.term_h + sp.term_tol);
                    float dh = 0.f;  // d(height reward)/d(height)
                    if (d >= -1.0f && d <= sp.term_tol) {
                        if (d < 0.0f) dh = -400.0f * d;
                        else if (d > 0.0f) dh = sp.h_scale;
                        else dh = 1.0f;
                    }
                    g_h += gr * dh;
                }
                v3 g_pos = mk3(0.f, g_h, 0.f);
                v3 g_w = ld3(go + 8);
                // lin_vel = vl - pos x w
                g_pos += cross(g_lv, w);
                g_w += cross(pos, g_lv);
                const q4 isr = mkq(sp.isr[0], sp.isr[1], sp.isr[2], sp.isr[3]);
                const q4 tq = qmul(r, isr);
                const v3 ey = mk3(0.f, 1.f, 0.f), ex_ = mk3(1.f, 0.f, 0.f);
                const v3 hd = rotate(tq, ex_);
                const float tx = sp.tgt_x - pos.x, tz = sp.tgt_z - pos.z;
                float l = sqrtf(tx * tx + tz * tz);
                const bool clamped = l < 1e-9f;
                l = clamped ? 1e-9f : l;
                const float dx = tx / l, dz = tz / l;
                q4 g_tq = rotate_adj_q(tq, ey, mk3(0.f, g_up, 0.f));
                g_tq += rotate_adj_q(tq, ex_, mk3(dx * g_hd, 0.f, dz * g_hd));
                q4 g_r = ldq(go + 1) + qmul_adj_a(isr, g_tq);
                if (!clamped) {
                    const float gdx = hd.x * g_hd, gdz = hd.z * g_hd;
                    const float dd = dx * gdx + dz * gdz;
                    g_pos.x -= (gdx - dx * dd) / l;
                    g_pos.z -= (gdz - dz * dd) / l;
                }
                add3(WF(aqn), g_pos);
                addq(WF(aqn) + 3, g_r);
                add3(WF(aqdn), g_w);
                add3(WF(aqdn) + 3, g_lv);
            }
        } else if (sp.kind == DSIM_ENV_PLANAR) {
            for (int k = lane; k < nq - 1; k += Exec::NL) {
                float g = go[k];
                if (sp.rew_kind == DSIM_REW_HOPPER) {
                    if (k == 0) {
                        const float d = q[1] - (sp.term_h + sp.term_tol);
                        float dh = 0.f;
                        if (d >= -1.0f && d <= 0.3f) dh = d < 0.0f ? -400.0f * d : (d > 0.0f ? sp.h_scale : 1.0f);
                        g += gr * dh;
                    } else if (k == 1) {
                        g += gr * (-2.0f * q[2] / (sp.pen[0] * sp.pen[0]));
                    }
                }
                WF(aqn)[1 + k] += g;
            }
            for (int k = lane; k < nd; k += Exec::NL) WF(aqdn)[k] += go[nq - 1 + k] + (k == 0 ? gr : 0.f);
            for (int k = lane; k < sp.n_act; k += Exec::NL) WF(gua)[k] = gr * sp.act_pen * 2.0f * WF(ua)[k];
        } else if (sp.kind == DSIM_ENV_CARTPOLE) {
            if (lane == 0) {
                const float th = atan2f(sinf(q[1]), cosf(q[1]));
                WF(aqn)[0] += go[0] - gr * 2.0f * q[0] * sp.pen[2];
                WF(aqdn)[0] += go[1] - gr * 2.0f * qd[0] * sp.pen[3];
                WF(aqn)[1] += go[2] * cosf(q[1]) - go[3] * sinf(q[1]) - gr * 2.0f * th * sp.pen[0];
                WF(aqdn)[1] += go[4] - gr * 2.0f * qd[1] * sp.pen[1];
                WF(gua)[0] = -gr * 2.0f * WF(ua)[0] * sp.act_pen;
            }
        }
    });
}

// whole fused env.step(): actions -> sim -> (q', qd', obs, rew) [+ episode bookkeeping: progress, done, restart]
template <class Ctx, class Exec>
DSIM_FN void dsim_env_fused_forward(const Ctx& c, Exec& ex, const DsimEnvSpec& sp, int substeps, int mm_freq,
                                    const float* g_q, const float* g_qd, const float* g_actions, float* g_q_out,
                                    float* g_qd_out, float* g_obs, float* g_rew, float* g_ckpt, const DsimEpisode& ep,
                                    int e, int n_envs, int* g_status = nullptr) {
    const int nq = c.d.nq, nd = c.d.nd;
    using IO = DsimIo<Ctx, Exec>;
    ex.begin_request();   // model constants first: memory returns loads in the order of their requests
    if constexpr (IO::PRE) {
        ex.fire([&](int lane) {   // early loads: requested now, used after the constants have arrived
            float* io = ex.io(lane);
            dsim_io_fetch<IO::CQ, Exec::NL>(io + IO::Q, g_q, nq, lane);
            dsim_io_fetch<IO::CD, Exec::NL>(io + IO::QD, g_qd, nd, lane);
            dsim_io_fetch<IO::CA, Exec::NL>(io + IO::ACT, g_actions, sp.n_act, lane);
            dsim_io_fetch<IO::CA, Exec::NL>(io + IO::SCALE, sp.act_scale, sp.n_act, lane);
        });
    }
    // requested last, needed last: these loads complete while the substeps run
    const long long p1 = ep.progress ? ep.progress[e] + 1 : 0;
    const int cnt = ep.progress ? ep.reset_count[e] : 0;
    ex.begin();
    dsim_init_static(c, ex);
    ex.run_both([&](int lane) { dsim_topo_init<false>(c, ex, lane); });   // every wave keeps its own topology records
    ex.run([&](int lane) {
        const float* io = ex.io(lane);
        dsim_io_each<IO::CQ, Exec::NL>(nq, lane, [&](int k, int u) { WF(q)[k] = IO::PRE ? io[IO::Q + u] : g_q[k]; });
        dsim_io_each<IO::CD, Exec::NL>(nd, lane, [&](int k, int u) { WF(qd)[k] = IO::PRE ? io[IO::QD + u] : g_qd[k]; });
        if (lane == 0) WF(epf)[0] = 0.f;
    });
    dsim_env_load_actions(c, ex, sp, g_actions, true);
    dsim_check_unit_quats(c, ex, g_status, e);
    for (int s = 0; s < substeps; ++s)
        dsim_fwd_substep(c, ex, (s % mm_freq) == 0, g_ckpt ? g_ckpt + (size_t)s * dsim_row(c) : nullptr,
                         g_ckpt ? dsim_ckpt_hinv(c, g_ckpt, substeps, s / mm_freq) : nullptr);
    float* tail = g_ckpt ? dsim_ckpt_tail(c, g_ckpt, substeps, mm_freq) : nullptr;
    if (!ep.progress) {
        ex.fire([&](int lane) {
            for (int k = lane; k < nq; k += Exec::NL) g_q_out[k] = WF(q)[k];
            for (int k = lane; k < nd; k += Exec::NL) g_qd_out[k] = WF(qd)[k];
            if (tail) {
                for (int k = lane; k < nq; k += Exec::NL) tail[k] = WF(q)[k];
                for (int k = lane; k < nd; k += Exec::NL) tail[nq + k] = WF(qd)[k];
                if (lane == 0) tail[nq + nd] = 0.f;
            }
        });
        dsim_env_observe(c, ex, sp, g_obs, g_rew);
        return;
    }
    dsim_env_obs_compute(c, ex, sp);
    if (ep.check_invalid) {
        // humanoid.py:340-356: non-finite observation / state, or |q|, |qd| > 1e6
        ex.run([&](int lane) {
            bool bad = false;
            for (int k = lane; k < nq; k += Exec::NL) bad = bad || !(fabsf(WF(q)[k]) <= 1e6f);
            for (int k = lane; k < nd; k += Exec::NL) bad = bad || !(fabsf(WF(qd)[k]) <= 1e6f);
            for (int k = lane; k < sp.n_obs; k += Exec::NL) bad = bad || !(fabsf(WF(obs)[k]) <= 3.4028235e38f);
            if (bad) WF(epf)[0] = 1.f;  // every writer stores the same value
        });
    }
    // `done` is assigned inside a main-wave phase: the helper wavefront (Exec::HAS_HELPER) keeps done == false and skips the
    // restart block below, which is correct ONLY because that block contains no workgroup barrier (plain run / fire phases).
    // A fork_join there would give the two waves different barrier counts and hang the GPU: derive `done` from values both
    // waves have (the flags in epf behind a run_both) before adding one.
    bool done = false;
    ex.run([&](int lane) {
        const float* o = WF(obs);
        const bool bad = ep.check_invalid && WF(epf)[0] != 0.f;
        done = (ep.height_terminate && o[0] < sp.term_h) || p1 > (long long)(ep.episode_length - 1) || bad;
        if (ep.obs_before)
            for (int k = lane; k < sp.n_obs; k += Exec::NL) ep.obs_before[(size_t)e * sp.n_obs + k] = o[k];
        if (tail) {
            for (int k = lane; k < nq; k += Exec::NL) tail[k] = WF(q)[k];
            for (int k = lane; k < nd; k += Exec::NL) tail[nq + k] = WF(qd)[k];
        }
        if (lane == 0) {
            g_rew[0] = bad ? 0.f : dsim_env_reward(c, sp);
            ep.progress[e] = done ? 0 : p1;
            ep.done[e] = done ? 1 : 0;
            if (tail) tail[nq + nd] = bad ? DSIM_EP_INVALID : (done ? DSIM_EP_DONE : 0.f);
            if (done) ep.reset_count[e] = cnt + 1;
        }
        if (!done) {
            for (int k = lane; k < sp.n_obs; k += Exec::NL) g_obs[k] = o[k];
            for (int k = lane; k < nq; k += Exec::NL) g_q_out[k] = WF(q)[k];
            for (int k = lane; k < nd; k += Exec::NL) g_qd_out[k] = WF(qd)[k];
        } else {
            // restart: every lane replaces exactly the words it has just read.  Start state = pool entry [+ fresh uniform
            // noise drawn here: coordinate k of q uses stream k, of qd stream 0x2000 + k, the root rotation stream 0x1000]
            const size_t slot = (size_t)(cnt % ep.pool) * n_envs + e;
            const bool root_rot = ep.noise_q && sp.kind == DSIM_ENV_LOCOMOTION && ep.noise_angle != 0.f;
            for (int k = lane; k < nq; k += Exec::NL) {
                float x = ep.reset_q[slot * nq + k];
                if (ep.noise_q) {
                    float u[4];
                    if (root_rot && k >= 3 && k < 7) {
                        // start rotation (x) rotation by (u0 - 0.5) * noise_angle about normalize((u1, u2, u3) - 0.5); all four
                        // lanes of the block evaluate the same quaternion and keep their own component
                        dsim_philox4(ep.seed, (unsigned)e, (unsigned)cnt, 0x1000u, u);
                        const float half = 0.5f * (u[0] - 0.5f) * ep.noise_angle;
                        v3 ax = mk3(u[1] - 0.5f, u[2] - 0.5f, u[3] - 0.5f);
                        float l = sqrtf(dot(ax, ax));
                        ax = ax * (1.0f / (l < 1e-9f ? 1e-9f : l));
                        const float sh = sinf(half), ch = cosf(half);
                        q4 dq = mkq(ax.x * sh, ax.y * sh, ax.z * sh, ch);
                        l = sqrtf(qdot(dq, dq));
                        dq = dq * (1.0f / (l < 1e-9f ? 1e-9f : l));
                        const q4 r0 = ldq(ep.reset_q + slot * nq + 3);
                        const q4 r = qmul(r0, dq);
                        x = k == 3 ? r.x : (k == 4 ? r.y : (k == 5 ? r.z : r.w));
                    } else {
                        dsim_philox4(ep.seed, (unsigned)e, (unsigned)cnt, (unsigned)k, u);
                        x += ep.noise_q[k] * (u[0] - 0.5f);
                    }
                }
                g_q_out[k] = WF(q)[k] = x;
            }
            for (int k = lane; k < nd; k += Exec::NL) {
                float x = ep.reset_qd[slot * nd + k];
                if (ep.noise_qd) {
                    float u[4];
                    dsim_philox4(ep.seed, (unsigned)e, (unsigned)cnt, 0x2000u + (unsigned)k, u);
                    x += ep.noise_qd[k] * (u[0] - 0.5f);
                }
                g_qd_out[k] = WF(qd)[k] = x;
            }
        }
    });
    if (done) {
        // observation of the new state with cleared stored actions (ant.py:228-233)
        ex.run([&](int lane) {
            for (int k = lane; k < sp.n_act; k += Exec::NL) WF(ua)[k] = 0.f;
        });
        dsim_env_obs_compute(c, ex, sp);
        ex.fire([&](int lane) {
            for (int k = lane; k < sp.n_obs; k += Exec::NL) g_obs[k] = WF(obs)[k];
        });
    }
}

// state-only observation (reset / initialize_trajectory path): obs of (q, qd) with the given stored actions
template <class Ctx, class Exec>
DSIM_FN void dsim_env_observe_only(const Ctx& c, Exec& ex, const DsimEnvSpec& sp, const float* g_q, const float* g_qd,
                                   const float* g_stored_actions, float* g_obs, float* g_rew) {
    ex.run([&](int lane) {
        for (int k = lane; k < c.d.nq; k += Exec::NL) WF(q)[k] = g_q[k];
        for (int k = lane; k < c.d.nd; k += Exec::NL) WF(qd)[k] = g_qd[k];
        for (int k = lane; k < sp.n_act; k += Exec::NL) WF(ua)[k] = g_stored_actions[k];
    });
    dsim_env_observe(c, ex, sp, g_obs, g_rew);
}

// reverse of dsim_env_fused_forward.  Cotangent inputs may be null (= zeros); the end state of the step and the episode
// flags come from the checkpoint tail.
template <class Ctx, class Exec>
DSIM_FN void dsim_env_fused_backward(const Ctx& c, Exec& ex, const DsimEnvSpec& sp, int substeps, int mm_freq,
                                     const float* g_ckpt, const float* g_actions, const float* g_gq_out,
                                     const float* g_gqd_out, const float* g_gobs, const float* g_grew,
                                     const float* g_gobs_before, float* g_gq_in, float* g_gqd_in, float* g_gactions) {
    const int nq = c.d.nq, nd = c.d.nd, M = c.d.M;
    ex.begin_request();   // model constants first: memory returns loads in the order of their requests
    const float* tail = dsim_ckpt_tail(c, const_cast<float*>(g_ckpt), substeps, mm_freq);
    const float ep_flags = tail[nq + nd];
    const bool live = ep_flags == 0.f;  // not restarted by this step: the returned state is the end state
    using IO = DsimIo<Ctx, Exec>;
    if constexpr (IO::PRE) {
        // early loads (requested before the flags above are looked at): end state, cotangents, actions, and the checkpoint
        // row of the last substep, which the first adjoint substep would otherwise wait for on the spot
        ex.fire([&](int lane) {
            float* io = ex.io(lane);
            dsim_io_fetch<IO::CQ, Exec::NL>(io + IO::Q, tail, nq, lane);
            dsim_io_fetch<IO::CD, Exec::NL>(io + IO::QD, tail + nq, nd, lane);
            dsim_io_fetch<IO::CQ, Exec::NL>(io + IO::GQ, g_gq_out, nq, lane);
            dsim_io_fetch<IO::CD, Exec::NL>(io + IO::GQD, g_gqd_out, nd, lane);
            dsim_io_fetch<IO::CA, Exec::NL>(io + IO::ACT, g_actions, sp.n_act, lane);
            dsim_io_fetch<IO::CA, Exec::NL>(io + IO::SCALE, sp.act_scale, sp.n_act, lane);
            dsim_io_fetch<IO::CO, Exec::NL>(io + IO::GOBS, g_gobs, sp.n_obs, lane);
            dsim_io_fetch<IO::CO, Exec::NL>(io + IO::GOBSB, g_gobs_before, sp.n_obs, lane);
            io[IO::GREW] = g_grew ? g_grew[0] : 0.f;
        });
        if constexpr (DsimHelperCommit<Ctx, Exec>::value) {
            ex.helper_prefetch(g_ckpt + (size_t)(substeps - 1) * dsim_row(c), dsim_row(c));
        } else {
            ex.prefetch(g_ckpt + (size_t)(substeps - 1) * dsim_row(c), dsim_row(c));
        }
    }
    ex.begin();
    if (ep_flags == DSIM_EP_INVALID) {
        // The step ended in a non-finite / exploded state: its reward was overwritten and its successor state replaced, so
        // only obs_before_reset could carry a cotangent -- through intermediates that are inf / NaN.  The reference
        // scrubs what comes out of that with nan_to_num hooks (humanoid.py:195-206); here the gradient is zero outright.
        ex.fire([&](int lane) {
            for (int k = lane; k < nq; k += Exec::NL) g_gq_in[k] = 0.f;
            for (int k = lane; k < nd; k += Exec::NL) g_gqd_in[k] = 0.f;
            for (int k = lane; k < sp.n_act; k += Exec::NL) g_gactions[k] = 0.f;
        });
        return;
    }
    // The inverse of the LAST group of substeps (the first the adjoint needs) is requested here and stored to LDS at the first
    // substep: its memory latency (measured: 2.4 k cycles of the Ant adjoint launch when loaded on the spot) passes under the
    // observation adjoint, whose serial quaternion chain on one lane needs no memory at all.
    const int groups = (substeps + mm_freq - 1) / mm_freq;
    // (helper-wave kernels: the helper brings it, if its registers hold it -- 4 x 64 x Exec::DSIM_APF words; a bigger model's main
    // wave loads it on the spot, as before)
    constexpr bool helper_aux = []() {
        if constexpr (DsimHelperCommit<Ctx, Exec>::value)
            return ((decltype(c.d)::nd * decltype(c.d)::nd + 3) & ~3) <= 4 * DSIM_NL * Exec::DSIM_APF;
        else return false;
    }();
    constexpr int HPF = []() {
        if constexpr (DsimHelperCommit<Ctx, Exec>::value) {
            return 0;   // (the helper wavefront brings it: helper_prefetch_aux below)
        } else if constexpr (Exec::WAVE_OPS) {
            // one-wave kernels without a helper run the launches beyond the helper capacity, where other waves cover the latency and
            // registers are the residency: Ant's env adjoint must stay within 256 VGPRs (two waves per SIMD; with these four and
            // the fused joint-space phase it was 258: 8192 environments 0.30 -> 0.49 ms)
            return 0;
        } else if constexpr (IO::PRE) {
            constexpr int need = (decltype(c.d)::nd * decltype(c.d)::nd + Exec::NL - 1) / Exec::NL;
            return need <= DSIM_HPF_MAX ? need : 0;
        } else {
            return 0;
        }
    }();
    if constexpr (HPF > 0) {
        const float* hv = dsim_ckpt_hinv(c, const_cast<float*>(g_ckpt), substeps, groups - 1);
        ex.fire([&](int lane) {
            float* hpf = ex.hpf(lane);
#pragma unroll
            for (int r = 0; r < HPF; ++r) hpf[r] = hv[(lane + Exec::NL * r) < nd * nd ? lane + Exec::NL * r : 0];
        });
    }
    // hinv <- a group's inverse, aH <- 0
    auto load_hinv = [&](int lane, const float* hv) __attribute__((always_inline)) {
        for (int k = lane; k < nd * nd; k += Exec::NL) {
            WF(hinv)[k] = hv[k];
            WF(aH)[k] = 0.f;
        }
    };
    dsim_init_static(c, ex);
    ex.run_both([&](int lane) { dsim_topo_init(c, ex, lane); });   // every wave keeps its own topology records
    // (helper wavefront) the inverse of the last group of substeps, the first the adjoint needs: requested now -- behind the
    // topology records, whose set-up needs the registers -- and stored by the helper at the first substep; the main wave's
    // observation adjoint in between covers the latency
    if constexpr (helper_aux)
        ex.helper_prefetch_aux(dsim_ckpt_hinv(c, const_cast<float*>(g_ckpt), substeps, groups - 1), dsim_hinv_words_d(nd));
    ex.run([&](int lane) {
        const float* io = ex.io(lane);
        dsim_io_each<IO::CQ, Exec::NL>(nq, lane, [&](int k, int u) {
            if constexpr (IO::PRE) {
                WF(q)[k] = io[IO::Q + u];
                WF(aqn)[k] = live ? io[IO::GQ + u] : 0.f;
            } else {
                WF(q)[k] = tail[k];
                WF(aqn)[k] = (live && g_gq_out) ? g_gq_out[k] : 0.f;
            }
        });
        dsim_io_each<IO::CD, Exec::NL>(nd, lane, [&](int k, int u) {
            if constexpr (IO::PRE) {
                WF(qd)[k] = io[IO::QD + u];
                WF(aqdn)[k] = live ? io[IO::GQD + u] : 0.f;
            } else {
                WF(qd)[k] = tail[nq + k];
                WF(aqdn)[k] = (live && g_gqd_out) ? g_gqd_out[k] : 0.f;
            }
            WF(aact)[k] = 0.f;
        });
        for (int k = lane; k < M; k += Exec::NL) WF(amact)[k] = 0.f;
    });
    dsim_env_load_actions(c, ex, sp, g_actions, true);
    dsim_env_observe_adjoint(c, ex, sp, g_gobs, g_grew, g_gobs_before, ep_flags, true);
    // The first inverse goes to LDS HERE, in front of the loops: a use inside them would keep the registers it waits in alive
    // through every substep (measured: Humanoid's helper kernel 255 -> 271 VGPRs).
    // (Kernels without a helper keep theirs in the loop: a phase of its own costs the four-wave SNUHumanoid kernel 2 %.)
    constexpr bool first_hinv_ahead = helper_aux;
    if constexpr (DsimHelperCommit<Ctx, Exec>::value) ex.group_sync();   // the main wave is done with the end state in q / qd (observation adjoint)
    if constexpr (helper_aux) {
        ex.helper_commit_aux(WF(hinv), WF(aH), dsim_hinv_words_d(nd));   // (published by the barrier of the first helper_commit)
        ex.fire([&](int lane) { dsim_hacc_zero(c, ex, lane); });
    }
    for (int g = groups - 1; g >= 0; --g) {
        const int s0 = g * mm_freq, s1 = (s0 + mm_freq < substeps) ? s0 + mm_freq : substeps;
        for (int s = s1 - 1; s >= s0; --s) {
            // forward intermediates of substep s (and, entering a group, the inverse its substeps used) from HBM.
            // The row was requested one substep earlier (ex.prefetch keeps it in flight in registers while the
            // previous adjoint substep computes), so this phase only moves registers to LDS.
            const float* hv = (s == s1 - 1 && !(first_hinv_ahead && g == groups - 1)) ? dsim_ckpt_hinv(c, const_cast<float*>(g_ckpt), substeps, g) : nullptr;
            if constexpr (DsimHelperCommit<Ctx, Exec>::value) {
                static_assert(IO::PRE, "helper-side commit belongs to the specialised kernels");
                ex.helper_commit(WF(q), dsim_row(c), s > 0 ? g_ckpt + (size_t)(s - 1) * dsim_row(c) : nullptr);
                if (hv)
                    ex.run([&](int lane) {
                        load_hinv(lane, hv);
                        dsim_hacc_zero(c, ex, lane);
                    });
            } else if (!DsimWideRows<Ctx, Exec>::value || s == substeps - 1) {
                // (DsimWideRows: only the launch's first row is committed here, every later one -- with its group's inverse -- by the
                // body level of the substep before it)
                if (!IO::PRE && s == substeps - 1) ex.prefetch(g_ckpt + (size_t)s * dsim_row(c), dsim_row(c));
                ex.run([&](int lane) {
                    ex.commit(WF(q), dsim_row(c), lane);
                    if (hv) {
                        if (HPF > 0 && g == groups - 1) {
                            const float* hpf = ex.hpf(lane);
#pragma unroll
                            for (int r = 0; r < HPF; ++r) {
                                const int k = lane + Exec::NL * r;
                                if (k < nd * nd) {
                                    WF(hinv)[k] = hpf[r];
                                    WF(aH)[k] = 0.f;
                                }
                            }
                        } else {
                            load_hinv(lane, hv);
                        }
                        dsim_hacc_zero(c, ex, lane);
                    }
                });
                if (s > 0) {
                    if constexpr (DsimWideRows<Ctx, Exec>::value) ex.prefetch_rest(g_ckpt + (size_t)(s - 1) * dsim_row(c), dsim_row(c));
                    else ex.prefetch(g_ckpt + (size_t)(s - 1) * dsim_row(c), dsim_row(c));
                }
            }
            if constexpr (Ctx::LEAN) dsim_bwd_recompute_forward(c, ex);
            if (s == s0) dsim_fwd_composite(c, ex);
            DsimNextRow nx;
            if constexpr (DsimWideRows<Ctx, Exec>::value) {
                nx.any = s > 0;
                nx.hv = (s == s0 && g > 0) ? dsim_ckpt_hinv(c, const_cast<float*>(g_ckpt), substeps, g - 1) : nullptr;
                nx.after = s > 1 ? g_ckpt + (size_t)(s - 2) * dsim_row(c) : nullptr;
            }
            dsim_bwd_substep(c, ex, s == s0, nx);  // aq / aqd (== aqn / aqdn, same LDS words) now belong to substep s - 1
        }
    }
    // sanitize: torch.nan_to_num(grad, 0.0, 0.0, 0.0) of the reference's per-step hooks on joint_q / joint_qd / actions
    // (envs/humanoid.py:195-206), applied where the three cotangents leave the launch
    const bool scrub = sp.sanitize != 0;
    auto clean = [&](float x) __attribute__((always_inline)) { return (scrub && !(fabsf(x) <= 3.4028235e38f)) ? 0.f : x; };
    ex.run([&](int lane) {
        for (int k = lane; k < nq; k += Exec::NL) g_gq_in[k] = clean(WF(aqn)[k]);
        for (int k = lane; k < nd; k += Exec::NL) g_gqd_in[k] = clean(WF(aqdn)[k]);
        const float* io = ex.io(lane);
        dsim_io_each<IO::CA, Exec::NL>(sp.n_act, lane, [&](int k, int u) {
            const float a = IO::PRE ? io[IO::ACT + u] : g_actions[k];
            const float sc = IO::PRE ? io[IO::SCALE + u] : sp.act_scale[k];
            float g = WF(gua)[k];
            if (sp.act_muscle) g = 0.5f * (g + sc * WF(amact)[k]);
            else g += sc * WF(aact)[sp.act_offset + k];
            g_gactions[k] = (a >= -1.0f && a <= 1.0f) ? clean(g) : 0.f;
        });
    });
}
