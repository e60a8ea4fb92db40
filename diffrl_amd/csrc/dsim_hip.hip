// dsim_hip.hip -- gfx950 kernels + the C ABI of include/dsim.h.
//
// Mapping: ONE environment per workgroup.  The phase code of an environment runs on ONE wavefront (64 lanes) -- plus, while
// all environments of a launch are resident at once, a HELPER wavefront that executes only the side blocks of split phases
// (DevExec<..., HELPER>) -- or on four wavefronts for models whose item lists are several wavefronts long (muscles).  The
// articulation template (a few KB) is copied from global memory into LDS once per launch, the environment's state row is
// loaded with one coalesced read per tensor, all `substeps` substeps run out of LDS, and only (q, qd) [+ the per-substep
// checkpoint when gradients are wanted] go back to HBM.  N=1024 environments put one main wave on each of the 1024 SIMDs
// of an MI355X; larger N stacks waves per SIMD and hides LDS/VALU latency.
//
// Every kernel exists in a GENERIC form (LDS offsets and model sizes are runtime values in SGPRs) and in
// per-model SPECIALISED forms (dsim_static_layouts.hpp: offsets are instruction immediates, sizes are
// compile-time loop bounds).  dsim_model_create picks a specialised form only if the layout it builds at
// run time is bit-identical to the generated table; the first profile of the generic kernels showed that
// >60 % of their instruction stream was SGPR spilling (v_readlane/v_writelane), address arithmetic and
// moves caused by ~100 runtime offsets (profiles/r01_*), which is what the specialisation removes.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC dsim_hip.hip -o libdsim_hip.so
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <new>
#include <string>

#define DSIM_FN __device__ __forceinline__
#ifndef DSIM_OPAQUE   // (-DDSIM_OPAQUE\(x\)= builds the A/B variant without it)
#define DSIM_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
#include "dsim_core.hpp"
#define DSIM_LIT_FN __device__ inline
#include "dsim_literal.hpp"
#ifdef DSIM_STATIC_LAYOUTS_FILE   // (python -m diffrl_amd.specialise --header-out: a generated header outside the tree)
#include DSIM_STATIC_LAYOUTS_FILE
#else
#include "dsim_static_layouts.hpp"
#endif

namespace {

// The workgroup IS one wavefront (64 threads, enforced at launch): a phase boundary needs no s_barrier and no wait at all.
// The DS instructions of one wave are executed by the LDS in issue order, so a load issued after a store observes it
// whichever lanes are involved; the only requirement is that the COMPILER does not move or cache LDS accesses across the
// boundary -- an empty asm with a memory clobber.  (Round 1 / early round 2 drained the queue with `s_waitcnt lgkmcnt(0)`
// here: every phase then paid the latency of its last store before the next phase's loads could even be issued; without
// it the results are bit-identical and Ant 1024 runs 1.7 % faster.  -DDSIM_WAVE_SYNC_ASM='"s_waitcnt lgkmcnt(0)"' builds
// the draining variant for A/B runs.)  Global loads / stores (vmcnt) are never waited for here either, which is what
// lets checkpoint traffic overlap with compute.
#ifndef DSIM_WAVE_SYNC_ASM
#define DSIM_WAVE_SYNC_ASM ""
#endif
__device__ __forceinline__ void dsim_wave_sync() { asm volatile(DSIM_WAVE_SYNC_ASM ::: "memory"); }

// reciprocal of a pivot: v_rcp_f32 + one Newton step (< 1 ulp on the pivots of an SPD matrix; ~3 instead of the ~11 instructions
// of the correctly rounded division on the pivot's dependent chain; -DDSIM_EXACT_DIV_SQRT builds the A/B variant with 1.0f / x)
__device__ __forceinline__ float dsim_pivot_rcp(float x) {
#if !defined(DSIM_EXACT_DIV_SQRT) && !defined(DSIM_EXACT_RCP)
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
#else
    return 1.0f / x;
#endif
}
// Gauss-Jordan inverse with lane i holding row i in registers.  Per pivot k the classical in-place update is
//   row k <- p = (row k with column k := 1) / h_kk;  row i <- (row i with column k := 0) - h_ik p        (dsim_core.hpp: dsim_fwd_mass)
// i.e. per column: broadcast h_kj, scale it, multiply-add, select the pivot row: four instructions (-DDSIM_CLASSIC_GJ builds that
// form for A/B runs; round 5 measured the deferred form below at -1 % .. -7 % of the forward launch, Ant .. SNUHumanoid).  The pivot
// row's scaling commutes with every LATER row operation (they are linear in the row), so lane k keeps its row UNSCALED -- column k
// set to 1 -- remembers 1 / h_kk and scales once at the end; every other row takes f h_kj with f = h_ik / h_kk, lane k with f = 0:
// per column one v_readlane and one multiply-add, no select.  Same elimination, different rounding (f h_kj instead of
// h_ik (h_kj / h_kk)); the generic kernels keep the classical form, the tests hold both to the reference.
template <int N, int NW> __device__ __forceinline__ void dsim_wave_gj(float* H) {
    const int lane = (int)threadIdx.x;
    if (NW > 1 && lane >= DSIM_NL) return;
    const int r = lane < N ? lane : 0;  // idle lanes shadow row 0; they store nothing
    float row[N];
#pragma unroll
    for (int j = 0; j < N; ++j) row[j] = H[r * N + j];
#ifndef DSIM_CLASSIC_GJ
    float scale = 1.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float piv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[k]), k));
        const float rp = dsim_pivot_rcp(piv);
        const bool own = lane == k;
        const float f = own ? 0.0f : row[k] * rp;
        scale = own ? rp : scale;
        row[k] = own ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float hkj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[j]), k));
            row[j] = __builtin_fmaf(-f, hkj, row[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) row[j] *= scale;
#else
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float piv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[k]), k));
        const float rp = dsim_pivot_rcp(piv);
        const float cik = row[k];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float hkj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[j]), k));
            const float pj = (j == k ? 1.0f : hkj) * rp;
            row[j] = (lane == k) ? pj : ((j == k ? 0.0f : row[j]) - cik * pj);
        }
    }
#endif
    if (lane < N) {
#pragma unroll
        for (int j = 0; j < N; ++j) H[lane * N + j] = row[j];
    }
}

// The same for a RUN-TIME size n <= NB (generic kernels: no compile-time nd): the matrix is padded with an identity block to
// NB x NB in registers -- lane i < n holds (row i, zeros), lane i >= n the unit row e_i -- so the padded pivots are 1, their
// multipliers 0, and the leading n x n block of the result is H^-1.  NB pivots of NB columns instead of 2 n phases of LDS round
// trips (round 5: the generic Ant forward launch spent 39 k of its 273 k cycles in those phases).
template <int NB, int NW> __device__ __forceinline__ void dsim_wave_gj_pad(float* H, int n) {
    const int lane = (int)threadIdx.x;
    if (NW > 1 && lane >= DSIM_NL) return;
    float row[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) row[j] = (lane < n && j < n) ? H[lane * n + j] : ((j == lane) ? 1.0f : 0.0f);
    float scale = 1.0f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const float piv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[k]), k));
        const float rp = dsim_pivot_rcp(piv);
        const bool own = lane == k;
        const float f = own ? 0.0f : row[k] * rp;
        scale = own ? rp : scale;
        row[k] = own ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float hkj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row[j]), k));
            row[j] = __builtin_fmaf(-f, hkj, row[j]);
        }
    }
    if (lane < n) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < n) H[lane * n + j] = row[j] * scale;
    }
}

// The same for two environments per wavefront (rows of the second one in lanes 32 ..): the pivot row travels through the LDS
// crossbar (ds_bpermute), whose source lane may differ between the halves.
template <int N> __device__ __forceinline__ void dsim_half_gj(float* H, int lane, int half_addr) {
    const int r = lane < N ? lane : 0;
    float row[N];
#pragma unroll
    for (int j = 0; j < N; ++j) row[j] = H[r * N + j];
#ifndef DSIM_CLASSIC_GJ
    float scale = 1.0f;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int src = (k << 2) + half_addr;
        const float piv = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row[k])));
        const float rp = dsim_pivot_rcp(piv);
        const bool own = lane == k;
        const float f = own ? 0.0f : row[k] * rp;
        scale = own ? rp : scale;
        row[k] = own ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float hkj = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row[j])));
            row[j] = __builtin_fmaf(-f, hkj, row[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) row[j] *= scale;
#else
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int src = (k << 2) + half_addr;
        const float piv = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row[k])));
        const float rp = dsim_pivot_rcp(piv);
        const float cik = row[k];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float hkj = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(row[j])));
            const float pj = (j == k ? 1.0f : hkj) * rp;
            row[j] = (lane == k) ? pj : ((j == k ? 0.0f : row[j]) - cik * pj);
        }
    }
#endif
    if (lane < N) {
#pragma unroll
        for (int j = 0; j < N; ++j) H[lane * N + j] = row[j];
    }
}

// The LDS image of one environment: [model constants | work arrays].  load(): the constants are copied from global memory
// (16 bytes per lane and load: const_words is a multiple of 4, both sides are 16-byte aligned) and the work area starts
// out as zeros, not as what the previous workgroup on this CU left behind: the weighted range sums (dsim_core.hpp:
// dsim_range_sum_m) multiply entries past the end of a range by 0, which needs them to be finite, and a few spare words
// (the "zero dof" behind atau) are constants 0 that no phase ever writes.
// request() / land(): the copy in two halves, so that the step functions can put their own early loads between them --
// memory returns loads in the order they were issued: constants first (needed first), then the step's inputs.  CW is the
// compile-time const_words of a specialised kernel (the constants wait in CW / 4 / lanes 16-byte registers) or 0 (generic
// kernels: request() does nothing, land() copies).
template <int NW, int CW, int EPW = 1> struct DsimImage {
    typedef float v4f __attribute__((ext_vector_type(4)));
    static constexpr int NLW = DSIM_NL * NW / EPW, REGS = (CW / 4 + NLW - 1) / NLW;
    // lane of this environment (EPW == 2: two environments per wavefront, 32 lanes each; `lds` then differs between the halves)
    static __device__ __forceinline__ int lane_() { return EPW == 1 ? (int)threadIdx.x : (int)(threadIdx.x & (NLW - 1)); }
    float* lds;    // where the constants go
    float* work;   // image base of the work arrays (== lds, except for the second environment of a pair)
    const uint32_t* cblob;
    int const_words, image_words;   // image_words: forward kernels fwd_words, adjoint kernels total_words
    v4f regs[REGS > 0 ? REGS : 1];
    __device__ __forceinline__ void request() {
        if constexpr (CW > 0) {
            const v4f* g = reinterpret_cast<const v4f*>(cblob);
#pragma unroll
            for (int r = 0; r < REGS; ++r) {
                const int i = lane_() + NLW * r;
                if (i < CW / 4) regs[r] = g[i];
            }
        }
    }
    __device__ __forceinline__ void land() {
        v4f* l = reinterpret_cast<v4f*>(lds);
        if constexpr (CW > 0) {
#pragma unroll
            for (int r = 0; r < REGS; ++r) {
                const int i = lane_() + NLW * r;
                if (i < CW / 4) l[i] = regs[r];
            }
        } else {
            const v4f* g = reinterpret_cast<const v4f*>(cblob);
            for (int i = lane_(); i < const_words / 4; i += NLW) l[i] = g[i];
        }
        v4f* w = reinterpret_cast<v4f*>(work);
        for (int i = const_words / 4 + lane_(); i < image_words / 4; i += NLW) w[i] = v4f{0.f, 0.f, 0.f, 0.f};
    }
};

// NW wavefronts per environment (workgroup of 64 * NW lanes).  NW == 1: a phase boundary is the compiler fence above;
// NW > 1: a workgroup barrier (phases whose item count exceeds 64 -- muscles, contacts, matrix entries of the bigger
// models -- are spread over the waves, which sit on different SIMDs of the CU).
// PF: 16-byte prefetch registers per lane for one checkpoint row (specialised kernels: exactly what the model's row needs)
// HELPER: a second wavefront per environment that executes nothing but the side tasks the phase code hands it with
// fork_join (contacts next to links in the kinematics, contacts^T next to the body-level cotangents, ...): those are
// blocks of a phase that run on OTHER LANES WITH DIFFERENT CODE than the phase's main block, which a single wavefront
// can only execute one after the other.  Unlike NW > 1 (every wave runs every phase on its share of the items) the
// helper skips all ordinary phases, so it adds almost nothing to the issue load of the SIMD it shares with another
// environment's main wave; a workgroup barrier of two waves costs ~30 cycles (measured), two per fork_join.
#ifdef DSIM_STAMPS
// developer builds only (tools/stamps.py): cycle stamps of workgroup 0's main wave at every phase boundary of the PRODUCT
// executor (helper wave and all), read back with dsim_debug_stamps.  clock64 is an SMEM read: it drains the wave's LDS queue.
__device__ long long g_dsim_stamps[2 * 16384];
#endif
// EPW == 2: TWO environments per wavefront, 32 lanes each (launches far beyond one wave per SIMD, where the SIMDs are saturated
// and every instruction of a wave whose phases use 9 .. 32 lanes is half wasted).  The phase code is the same: it sees NL = 32
// lanes and an LDS image pointer that differs between the halves; the cross-lane primitives stay inside the half.
template <int NW, int PF = 6, int CW = 0, bool HELPER = false, int EPW = 1> struct DevExec {
    static_assert(EPW == 1 || (EPW == 2 && NW == 1 && !HELPER), "two environments per wavefront: one-wave mapping, no helper");
    static constexpr int ENVS_PER_WAVE = EPW;
#ifdef DSIM_STAMPS
    int stamp_i_ = 0, stamp_tag_ = 0;
    // main wave: entries [0, 8192), helper wave: [8192, 16384) (its tags + 50)
    __device__ __forceinline__ void stamp() {
        // (several wavefronts per environment: the second track is the SECOND wavefront's first lane)
        const bool main_lane = threadIdx.x == 0, help_lane = (HELPER || NW > 1) && threadIdx.x == DSIM_NL;
        if (blockIdx.x == 0 && (main_lane || help_lane) && stamp_i_ < 8192) {
            const int k = stamp_i_ + (help_lane ? 8192 : 0);
            g_dsim_stamps[k] = clock64();
            g_dsim_stamps[16384 + k] = stamp_tag_ + (help_lane ? 50 : 0);
        }
        ++stamp_i_;
        ++stamp_tag_;
    }
    __device__ __forceinline__ void mark(int t) { stamp_tag_ = t * 100; }
#else
    __device__ __forceinline__ void stamp() {}
    __device__ __forceinline__ void mark(int) {}
#endif
    static_assert(!HELPER || NW == 1, "the helper wavefront belongs to the one-wave mapping");
    static constexpr bool HAS_HELPER = HELPER;
    const bool helper_ = HELPER && threadIdx.x >= DSIM_NL;
    __device__ __forceinline__ int lane_() const { return (HELPER || EPW > 1) ? (int)(threadIdx.x & (NL - 1)) : (int)threadIdx.x; }
    // first lane of this environment's half of the wave (EPW == 2), as a ds_bpermute byte address
    const int half_addr_ = EPW > 1 ? (int)(threadIdx.x & (DSIM_NL / EPW)) << 2 : 0;
    // both waves: LDS operations of this wave done, workgroup barrier (no wait for global memory)
    __device__ __forceinline__ void group_barrier() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    static constexpr int NL = DSIM_NL * NW / EPW;
    static constexpr int DSIM_PF = PF;
    // Cross-lane primitives of the wavefront (one wave per environment only): the phase code uses them for the small
    // point-to-point exchanges between lanes that would otherwise be an LDS store, a phase boundary and an LDS load.
    static constexpr bool WAVE_OPS = NW == 1;
    // ... and inside run_wave0 (a phase that only the FIRST wavefront of the environment executes) with any number of waves
    static constexpr bool WAVE0_OPS = true;
    // value of v in lane `src` (any lane): ds_bpermute_b32 -- the LDS crossbar, no LDS memory, no phase boundary.  All
    // lanes of the wave must execute it (uniform control flow); values of lanes that hold nothing meaningful are ignored.
    __device__ __forceinline__ float shfl(float v, int src) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute((src << 2) + half_addr_, __float_as_int(v)));
    }
    // value of v in lane `src`, src uniform across the wave (a compile-time constant after unrolling): v_readlane_b32
    // (two environments per wave: the source differs between the halves -- the crossbar again)
    __device__ __forceinline__ float bcast(float v, int src) {
        if constexpr (EPW > 1) return shfl(v, src);
        else return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
    }
    // value of v in lane + D of the same 16-lane ROW (0 beyond the row's end, 0 from an inactive lane): a DPP row shift on the
    // operand -- VALU latency, no LDS, no SGPR.  Pinned on hardware by tools/micro/dpp_semantics.hip (row_shl:D, bound_ctrl on).
    template <int D> __device__ __forceinline__ float from_above(float v) {
        static_assert(D >= 1 && D <= 15, "row shifts reach 1 .. 15 lanes");
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + D, 0xf, 0xf, true));
    }
    // ... and lane - D of the row (row_shr:D)
    template <int D> __device__ __forceinline__ float from_below(float v) {
        static_assert(D >= 1 && D <= 15, "row shifts reach 1 .. 15 lanes");
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, true));
    }
    // a_k += a_k[lane + D of the row] * w for six values at once: v_fmac_f32 with the DPP shift ON ITS OPERAND (the compiler
    // keeps a v_mov_b32_dpp + v_fmac pair for the builtin form: twice the instructions on a path where every issue slot counts).
    // A VGPR written by a VALU instruction may be read through DPP two wait states later at the earliest: the six lines are
    // independent of each other, so only the first needs the s_nop, and a_k's own update is five instructions behind its read.
    // FIRST: the six values may have been written by the instructions right before (the s_nop); later steps of a sum read what
    // the previous step's block wrote five instructions earlier.
    template <int D, bool FIRST = true>
    __device__ __forceinline__ void add_from_above(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        static_assert(D >= 1 && D <= 15, "row shifts reach 1 .. 15 lanes");
        if constexpr (FIRST) asm volatile("s_nop 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
#ifdef DSIM_NO_DPP_ASM   // (A/B builds: the builtin form)
        a0 = __builtin_fmaf(from_above<D>(a0), w, a0); a1 = __builtin_fmaf(from_above<D>(a1), w, a1);
        a2 = __builtin_fmaf(from_above<D>(a2), w, a2); a3 = __builtin_fmaf(from_above<D>(a3), w, a3);
        a4 = __builtin_fmaf(from_above<D>(a4), w, a4); a5 = __builtin_fmaf(from_above<D>(a5), w, a5);
#else
        asm volatile("v_fmac_f32_dpp %0, %0, %6 row_shl:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %6 row_shl:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %2, %2, %6 row_shl:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %3, %3, %6 row_shl:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %4, %4, %6 row_shl:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %5, %5, %6 row_shl:%7 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5)
                     : "v"(w), "n"(D));
#endif
    }
    // every LDS load issued so far has landed: ONE s_waitcnt lgkmcnt(0) behind a batch of loads instead of the partial waits
    // (lgkmcnt(n), n counting down) the compiler otherwise puts in front of each first use.  A lone wave pays a full issue slot
    // (~4.6 cycles, tools/micro/issue_mix.hip) for every s_waitcnt, satisfied or not, while the loads of a batch return a few
    // cycles apart: after the phase's load batch the one full wait is cheaper than a dozen partial ones.
    __device__ __forceinline__ void loads_landed() {
#ifndef DSIM_NO_FULL_WAIT   // (A/B builds)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // vmcnt 63, expcnt 7, lgkmcnt 0
#endif
    }
    // the same from the NEXT lane of the wave (wave_shl:1: crosses the 16-lane row boundaries; 0 into lane 63)
    template <bool FIRST = true>
    __device__ __forceinline__ void add_from_next(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        if constexpr (FIRST) asm volatile("s_nop 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
        asm volatile("v_fmac_f32_dpp %0, %0, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %2, %2, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %3, %3, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %4, %4, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %5, %5, %6 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5)
                     : "v"(w));
    }
    // ... and from ONE given lane SRC (an edge of the tree that no shift reaches): v_readlane into an SGPR, weighted add
    template <int SRC>
    __device__ __forceinline__ void add_from_lane(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        const float s0 = bcast(a0, SRC), s1 = bcast(a1, SRC), s2 = bcast(a2, SRC), s3 = bcast(a3, SRC), s4 = bcast(a4, SRC), s5 = bcast(a5, SRC);
        a0 = __builtin_fmaf(s0, w, a0); a1 = __builtin_fmaf(s1, w, a1); a2 = __builtin_fmaf(s2, w, a2);
        a3 = __builtin_fmaf(s3, w, a3); a4 = __builtin_fmaf(s4, w, a4); a5 = __builtin_fmaf(s5, w, a5);
    }
    // this wave's earlier LDS stores are visible to its later LDS loads (DS operations of a wave execute in order)
    __device__ __forceinline__ void lds_fence() { dsim_wave_sync(); }
    // earlier stores to (host-mapped) global memory are visible system-wide before later ones (the model's status words)
    __device__ __forceinline__ void system_fence() { __threadfence_system(); }
    __device__ __forceinline__ void sync() {
        if constexpr (NW == 1) dsim_wave_sync();
        else __syncthreads();
        stamp();
    }
    template <class F> __device__ __forceinline__ void run(F&& f) {
        if (!helper_) f(lane_());
        sync();
    }
    // phase of the first wavefront alone: the link-lane phases of a model whose per-item phases need several wavefronts (row-tree
    // body level of the muscle model).  Inside, the cross-lane primitives above are those of that one wave; side_done() is a no-op
    // (everything the phase reads was finished behind the workgroup barrier in front of it).
    template <class F> __device__ __forceinline__ void run_wave0(F&& f) {
        if constexpr (NW == 1) {
            run(f);
        } else {
            if (threadIdx.x < DSIM_NL) f((int)threadIdx.x);
            sync();
        }
    }
    // phase executed by the helper as well (register-only set-up such as the topology records: each wave keeps its own)
    template <class F> __device__ __forceinline__ void run_both(F&& f) {
        f(lane_());
        sync();
    }
    // fm and fh are two blocks of one phase that touch disjoint LDS words.  With a helper: barrier (everything earlier is
    // visible to both waves), main wave fm / helper wave fh, barrier.  Without: one after the other, as a single wave must.
    template <class FM, class FH> __device__ __forceinline__ void fork_join(FM&& fm, FH&& fh) {
        if constexpr (HELPER) {
            group_barrier();
            if (helper_) fh(lane_());
            else fm(lane_());
            stamp();          // (this wave's block is done; the next stamp is behind the barrier: the difference is waiting)
            group_barrier();
            stamp();
        } else {
            fm(lane_());
            fh(lane_());
            sync();
        }
    }
    // The same with a hand-over INSIDE the main block: fm calls mid() exactly once, at the point from which fh may read what
    // fm has written so far (the kinematics publish poses and twists, then go on with inertias and body forces while the
    // helper evaluates the contacts).  The helper waits at mid()'s barrier instead of at a barrier before the phase; what fm
    // keeps in registers across mid() stays there.  Without a helper: fm, then fh.
    template <class FM, class FH> __device__ __forceinline__ void fork_join_mid(FM&& fm, FH&& fh) {
        if constexpr (HELPER) {
            group_barrier();   // the helper is done with whatever it was left to do beside the previous phase (fork_mid_detached)
            if (helper_) {
                group_barrier();
                fh(lane_());
            } else {
                fm(lane_());
            }
            stamp();
            group_barrier();
            stamp();
        } else {
            fm(lane_());
            fh(lane_());
            sync();
        }
    }
    __device__ __forceinline__ void mid() {
        if constexpr (HELPER || NW > 1) group_barrier();
        else dsim_wave_sync();
        if constexpr (NW > 1) stamp();
    }
    // Several wavefronts per environment (dsim_core.hpp: DsimWideOverlap): f0 is the block of the FIRST wavefront (link lanes, with
    // its cross-lane primitives), fr the block of the others, which sees lanes 0 .. NL - 64 - 1 -- two different instruction
    // streams at the same time.  Where one needs what the other has written both call the same hand-over point -- mid(), mid2(),
    // side_done(): workgroup barriers, so every wavefront must pass each of them exactly once -- and the phase ends with the
    // barrier of sync().  (A branch on the wavefront index is uniform per wave: s_barrier counts waves, not lanes.)
    template <class F0, class FR> __device__ __forceinline__ void fork_wave0(F0&& f0, FR&& fr) {
        static_assert(NW > 1, "fork_wave0 belongs to the mapping with several wavefronts per environment");
        if (threadIdx.x < DSIM_NL) f0((int)threadIdx.x);
        else fr((int)threadIdx.x - DSIM_NL);
        asm volatile("" ::: "memory");
        stamp();
        group_barrier();
        stamp();
    }
    __device__ __forceinline__ void mid2() {
        if constexpr (NW > 1) {
            stamp();   // (developer builds: this wave's arrival; the next stamp is behind the barrier -- the difference is waiting)
            group_barrier();
            stamp();
        }
    }
    // ... and without the barrier at the end: fh is a DETACHED side block -- nothing it reads is written, and nothing it writes is
    // read, before the next barrier that both waves take (the checkpoint copy beside the integrator: the next substep's
    // kinematics start with one).  The main wave goes on at once.
    template <class FM, class FH> __device__ __forceinline__ void fork_mid_detached(FM&& fm, FH&& fh) {
        if constexpr (HELPER) {
            if (helper_) {
                group_barrier();
                fh(lane_());
            } else {
                fm(lane_());
            }
            asm volatile("" ::: "memory");
            stamp();
        } else {
            fm(lane_());
            fh(lane_());
            sync();
        }
    }
    // fh is a side block whose RESULTS fm needs part-way through: fm calls side_done() exactly once, at the point from which it
    // reads what fh wrote (the row-tree body level: the link lanes pick up the per-body contact sums after the first third of
    // their work).  With a helper: the helper runs fh and arrives at the barrier that side_done() is for the main wave; no barrier
    // before (fh must read nothing the main wave wrote since the last barrier both waves took) and none after (the helper is
    // idle until the next fork).  Without: fh, then fm.
    template <class FM, class FH> __device__ __forceinline__ void fork_side(FM&& fm, FH&& fh) {
        if constexpr (HELPER) {
            if (helper_) {
                fh(lane_());
                group_barrier();
            } else {
                fm(lane_());
            }
            asm volatile("" ::: "memory");
            stamp();
        } else {
            fh(lane_());
            dsim_wave_sync();
            fm(lane_());
            sync();
        }
    }
    // (several wavefronts: a no-op inside run_wave0, whose inputs were finished behind the barrier in front of it; the barrier of
    // fork_wave0's first wavefront -- side_done_w())
    __device__ __forceinline__ void side_done() {
        if constexpr (HELPER) group_barrier();
        else if constexpr (NW == 1) dsim_wave_sync();
    }
    __device__ __forceinline__ void side_done_w() {
        stamp();
        group_barrier();
        stamp();
    }
    // phase that only writes global memory nobody in this launch reads back: no vmcnt wait.  One wave: no barrier either
    // (its LDS reads precede, in program order, whatever the next phase stores); several waves: the LDS words it reads
    // must not be overwritten by a wave that runs ahead, hence a barrier.
    template <class F> __device__ __forceinline__ void fire(F&& f) {
        if (!helper_) f(lane_());
        if constexpr (NW > 1) __syncthreads();
    }
    // begin(): the model constants arrive in LDS and the work area is cleared (see DsimImage).  The step functions call it
    // AFTER they have requested their own inputs from global memory (dsim_core.hpp: early loads), so that the launch
    // pays ONE memory latency for all of them instead of one per prologue phase.
    DsimImage<NW, CW, EPW> img_;
    __device__ __forceinline__ void begin_request() {
        if (!helper_) img_.request();
    }
    __device__ __forceinline__ void begin() {
        if (!helper_) img_.land();
        if constexpr (HELPER) group_barrier();
        else sync();
    }
    // per-lane registers of the early loads
    float io_[DSIM_IO_MAX];
    __device__ __forceinline__ float* io(int) { return io_; }
    // Gauss-Jordan inverse of the N x N matrix at H (LDS, row-major), in place, by the first wavefront: lane i holds row i
    // in registers, the pivot row travels through v_readlane (dsim_core.hpp: dsim_fwd_mass has the formulas).
    template <int N> __device__ __forceinline__ void wave_gj(float* H) {
        if constexpr (EPW > 1) dsim_half_gj<N>(H, lane_(), half_addr_);
        else if (!helper_) dsim_wave_gj<N, NW>(H);
        sync();
    }
    // ... of a run-time size n <= NB (generic kernels), padded in registers
    static constexpr bool WAVE_GJ_PAD = EPW == 1;
    template <int NB> __device__ __forceinline__ void wave_gj_pad(float* H, int n) {
        if (!helper_) dsim_wave_gj_pad<NB, NW>(H, n);
        sync();
    }
    // lane-private accumulators of the mass-matrix cotangent (dsim_core.hpp: DSIM_HACC_MAX registers per lane)
    float hacc_[DSIM_HACC_MAX];
    __device__ __forceinline__ float* hacc(int) { return hacc_; }
    float hpf_[DSIM_HPF_MAX];
    __device__ __forceinline__ float* hpf(int) { return hpf_; }
    // per-lane topology records (dsim_core.hpp: DsimTopoRegs, dsim_topo_init)
    DsimTopoRegs topo_;
    __device__ __forceinline__ DsimTopoRegs& topo(int) { return topo_; }
    // software prefetch of one checkpoint row: global loads are issued here and stay in flight (registers) until
    // commit() stores them to LDS one adjoint substep later; rows longer than 64*DSIM_PF lanes*regs are read at commit
    // (a native vector type: the may_alias struct dsim_f4 of the copy loops is not promoted to registers -- an array of
    // it ended up in scratch memory with an s_waitcnt vmcnt(0) right behind every prefetch load)
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f pf[DSIM_PF];
    const float* pf_src;
    __device__ __forceinline__ void prefetch(const float* row, int words) {
        pf_src = row;
        if (helper_ || words > 4 * NL * DSIM_PF) return;
        const v4f* r4 = reinterpret_cast<const v4f*>(row);
#pragma unroll
        for (int r = 0; r < DSIM_PF; ++r) {
            const int k = lane_() + NL * r;
            if (4 * k < words) pf[r] = r4[k];
        }
    }
    // ---- the same on the HELPER wavefront (dsim_core.hpp: DsimHelperCommit) ------------------------------------------------
    // both waves: workgroup barrier (the helper waits for the main wave's prologue)
    __device__ __forceinline__ void group_sync() {
        if constexpr (HELPER) group_barrier();
    }
    __device__ __forceinline__ void helper_prefetch(const float* row, int words) {
        if constexpr (HELPER) {
            pf_src = row;
            if (!helper_ || words > 4 * NL * DSIM_PF) return;
            const v4f* r4 = reinterpret_cast<const v4f*>(row);
#pragma unroll
            for (int r = 0; r < DSIM_PF; ++r) {
                const int k = lane_() + NL * r;
                if (4 * k < words) pf[r] = r4[k];
            }
        }
    }
    // helper: the prefetched row -> LDS, then the request for the row after it (next != nullptr); both waves: barrier
    __device__ __forceinline__ void helper_commit(float* dst, int words, const float* next) {
        if constexpr (HELPER) {
            if (helper_) {
                commit(dst, words, lane_());
                if (next) helper_prefetch(next, words);
            }
            asm volatile("" ::: "memory");
            group_barrier();
            stamp();
        }
    }
    // A second block in flight on the helper: the inverse of the mass matrix that the adjoint's first substep needs (fused env
    // adjoint).  Requested at kernel start next to the first row, stored (and its cotangent zeroed) in front of the barrier of the
    // first helper_commit -- the main wave never touches it (Humanoid: 729 words, 12 per lane, that the main wave has no
    // registers to keep in flight across its prologue).  `words`: a multiple of 4, at most 4 * 64 * DSIM_APF.
    static constexpr int DSIM_APF = 4;
    v4f apf[DSIM_APF];
    __device__ __forceinline__ void helper_prefetch_aux(const float* src, int words) {
        if constexpr (HELPER) {
            if (!helper_) return;
            const v4f* s4 = reinterpret_cast<const v4f*>(src);
#pragma unroll
            for (int r = 0; r < DSIM_APF; ++r) {
                const int k = lane_() + NL * r;
                if (4 * k < words) apf[r] = s4[k];
            }
        }
    }
    __device__ __forceinline__ void helper_commit_aux(float* dst, float* zeroed, int words) {
        if constexpr (HELPER) {
            if (!helper_) return;
            v4f *d4 = reinterpret_cast<v4f*>(dst), *z4 = reinterpret_cast<v4f*>(zeroed);
#pragma unroll
            for (int r = 0; r < DSIM_APF; ++r) {
                const int k = lane_() + NL * r;
                if (4 * k < words) {
                    d4[k] = apf[r];
                    z4[k] = v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    // the same by the wavefronts BEHIND the first one alone (dsim_core.hpp: DsimWideRows; NL - 64 lanes): requested from any phase
    // (the first wavefront returns at once), committed from fork_wave0's second block, which sees lanes 0 .. NL - 64 - 1
    __device__ __forceinline__ void prefetch_rest(const float* row, int words) {
        constexpr int NLR = NL - DSIM_NL;
        pf_src = row;
        if (threadIdx.x < DSIM_NL || words > 4 * NLR * DSIM_PF) return;
        const v4f* r4 = reinterpret_cast<const v4f*>(row);
#pragma unroll
        for (int r = 0; r < DSIM_PF; ++r) {
            const int k = (int)threadIdx.x - DSIM_NL + NLR * r;
            if (4 * k < words) pf[r] = r4[k];
        }
    }
    __device__ __forceinline__ void commit_rest(float* dst, int words, int lane) {
        constexpr int NLR = NL - DSIM_NL;
        if (words > 4 * NLR * DSIM_PF) {
            for (int k = lane; k < words; k += NLR) dst[k] = pf_src[k];
            return;
        }
        v4f* d4 = reinterpret_cast<v4f*>(dst);
#pragma unroll
        for (int r = 0; r < DSIM_PF; ++r) {
            const int k = lane + NLR * r;
            if (4 * k < words) d4[k] = pf[r];
        }
    }
    __device__ __forceinline__ void commit(float* dst, int words, int lane) {
        if (words > 4 * NL * DSIM_PF) {
            for (int k = lane; k < words; k += NL) dst[k] = pf_src[k];
            return;
        }
        v4f* d4 = reinterpret_cast<v4f*>(dst);
#pragma unroll
        for (int r = 0; r < DSIM_PF; ++r) {
            const int k = lane + NL * r;
            if (4 * k < words) d4[k] = pf[r];
        }
    }
};

// Launch modes of the step kernels: 0 one environment per workgroup of NW waves, 1 ... plus its helper wavefront, 2 TWO
// environments per wavefront (DevExec EPW == 2): environment 2 b + h in half h of workgroup b's only wave, LDS images back to back
// Minimum resident waves per SIMD the forward pair kernels are compiled for (0: no constraint).  Ant's env-step pair kernel
// needs 173 VGPRs, five over the 168 of three waves per SIMD: compiled for three it spills those to scratch and the Ant 8192
// rollout drops from 18.4 M to 17.4 M env-steps/s (profiles/r04_pair_ab.txt), so two waves per SIMD (16 environments per CU,
// plain mapping: 12) it is.
#ifndef DSIM_PAIR_WAVES
#define DSIM_PAIR_WAVES 0
#endif
// Kernels with several wavefronts per environment (SNUHumanoid: 4 waves, LDS for two environments per CU) need two resident waves
// per SIMD -- 256 registers per lane, accumulation registers included: one more halves the environments in flight and doubles the
// launch time (measured in round 5: the operator-level adjoint at 257: 0.29 -> 0.53 ms).  The bound makes the compiler keep to it.
// (Not the lean-checkpoint kernels: their adjoint carries the forward phases too and would spill to scratch memory under it.)
#define DSIM_WIDE_WAVES(NW) (((NW) > 1 && !LEAN) ? 2 : 0)
#define DSIM_MODE_PLAIN 0
#define DSIM_MODE_HELPER 1
#define DSIM_MODE_PAIR 2
template <class O, class D> struct KCommonT {
    O o;
    D d;
    const uint32_t* cblob;
    float h;
    int substeps, mm_freq, n_envs;
    long long ckpt_stride;  // floats per environment (dsim_ckpt_words)
    int* status;            // the model's two status words (host memory mapped into the device): dsim_check_unit_quats
};

// prefetch registers a model's checkpoint row needs (compile-time layouts), or the generic default of 6 (rows up to 1536 floats
// at one wavefront per environment; longer rows are read at commit time)
// models that get a helper wavefront: specialised one-wave kernels of models with ground contacts and without muscles
// FWD: a forward kernel.  The GENERIC one-wave kernels (run-time layout) get a helper wavefront for their ADJOINT only (round 5):
// its fork_join phases -- the af block beside the accumulators of adj H and the per-dof cotangents, contacts^T beside the per-link
// block of the body level -- are the same phase code; the generic forward has no split phase, a helper would only idle there.
template <class D, int NW, bool FWD = false> constexpr bool dsim_has_helper() {
#ifdef DSIM_NO_HELPER
    return false;
#else
    if constexpr (std::is_empty<D>::value) return NW == 1 && D::C > 0 && D::NS == 0 && D::L + D::C <= DSIM_NL;
#ifdef DSIM_NO_GENERIC_HELPER   // (A/B builds)
    else return false;
#else
    else return NW == 1 && !FWD;
#endif
#endif
}
// models that get two-environments-per-wave kernels: specialised one-wave models whose phases fit 32 lanes
template <class D, int NW> constexpr bool dsim_has_pair() {
#ifdef DSIM_NO_PAIR
    return false;
#else
    return NW == 1 && dsim_pair_ok<D>();
#endif
}
// f(lean, mode) with the two launch-time choices as compile-time constants.  `help` is false for models without helper
// kernels, so they instantiate nothing extra.
// FWD: a forward kernel.  Pair kernels exist for the forward direction only: the adjoint's LDS image (Ant 17 KB) caps the
// environments per CU at the same 8 with either mapping, and a pair wave is 1.26 x as long as a plain one (measured,
// profiles/r04_pair_ab.txt: Ant 8192 adjoint 0.287 ms plain, 0.349 ms pair; -DDSIM_PAIR_BWD builds them for A/B runs).
template <class D, int NW, bool FWD, class F> int dsim_with_flags(bool lean, int mode, F&& f) {
    using M0 = std::integral_constant<int, DSIM_MODE_PLAIN>;
    using M1 = std::integral_constant<int, DSIM_MODE_HELPER>;
    using M2 = std::integral_constant<int, DSIM_MODE_PAIR>;
    if constexpr (dsim_has_helper<D, NW, FWD>()) {
        if (mode == DSIM_MODE_HELPER) return lean ? f(std::true_type{}, M1{}) : f(std::false_type{}, M1{});
    }
#ifdef DSIM_PAIR_BWD
    constexpr bool pair_dir = true;
#else
    constexpr bool pair_dir = FWD;
#endif
    if constexpr (dsim_has_pair<D, NW>() && pair_dir) {
        if (mode == DSIM_MODE_PAIR) return lean ? f(std::true_type{}, M2{}) : f(std::false_type{}, M2{});
    }
    return lean ? f(std::true_type{}, M0{}) : f(std::false_type{}, M0{});
}
template <class O> constexpr int dsim_const_words() {
    if constexpr (std::is_empty<O>::value) return O::const_words;
    else return 0;
}
template <class O, int NW, bool LEAN, int MODE = 0> constexpr int dsim_pf_regs() {
    if constexpr (std::is_empty<O>::value) {
        // (several wavefronts: rows may be brought in by the wavefronts behind the first one alone, DevExec::prefetch_rest)
        constexpr int words = LEAN ? O::xsc - O::q : O::save_words, lanes = DSIM_NL * (NW > 1 ? NW - 1 : NW) / (MODE == 2 ? 2 : 1);
        return (words / 4 + lanes - 1) / lanes;
    } else {
        return 6;
    }
}
template <int MODE> __device__ __forceinline__ int dsim_env_index() {
    return MODE == DSIM_MODE_PAIR ? 2 * (int)blockIdx.x + (int)(threadIdx.x >> 5) : (int)blockIdx.x;
}
// context of this workgroup's environment; the executor gets the image description and loads it in begin().
// A pair shares ONE copy of the model constants (Ant: 1096 of the forward image's 2332 words; each half loads it, same values
// to the same words, so that a half without an environment -- odd n_envs -- may leave at once): [constants | work 0 | work 1].
template <bool LEAN, int MODE, class Exec, class O, class D>
__device__ __forceinline__ DsimCtxT<O, D, LEAN> start_env(float* lds, const KCommonT<O, D>& k, int image_words, Exec& ex) {
    float* base = MODE == DSIM_MODE_PAIR ? lds + (threadIdx.x >> 5) * (image_words - k.o.const_words) : lds;
    ex.img_.lds = lds;
    ex.img_.work = base;
    ex.img_.cblob = k.cblob;
    ex.img_.const_words = k.o.const_words;
    ex.img_.image_words = image_words;
    DsimCtxT<O, D, LEAN> c;
    c.s = base;
    c.k = lds;
    c.o = k.o;
    c.d = k.d;
    c.h = k.h;
    return c;
}

template <class O, class D, int NW, bool LEAN, int MODE>
__global__ __launch_bounds__((DSIM_NL * NW * (MODE == 1 ? 2 : 1)), (MODE == DSIM_MODE_PAIR ? DSIM_PAIR_WAVES : DSIM_WIDE_WAVES(NW))) void dsim_fwd_kernel(KCommonT<O, D> k, const float* __restrict__ q_in,
                                                           const float* __restrict__ qd_in,
                                                           const float* __restrict__ act,
                                                           const float* __restrict__ mact, float* q_out,
                                                           float* qd_out, float* ckpt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = dsim_env_index<MODE>();
    if (e >= k.n_envs) return;
    DevExec<NW, dsim_pf_regs<O, NW, LEAN, MODE>(), dsim_const_words<O>(), MODE == 1, MODE == 2 ? 2 : 1> ex;
    auto c = start_env<LEAN, MODE>(lds, k, k.o.fwd_words, ex);
    const size_t nq = k.d.nq, nd = k.d.nd, M = k.d.M;
    dsim_sim_step_forward(c, ex, k.substeps, k.mm_freq, q_in + e * nq, qd_in + e * nd, act + e * nd,
                          M ? mact + e * M : nullptr, q_out + e * nq, qd_out + e * nd,
                          ckpt ? ckpt + (size_t)e * k.ckpt_stride : nullptr, k.status, e);
}

template <class O, class D, int NW, bool LEAN, int MODE>
__global__ __launch_bounds__((DSIM_NL * NW * (MODE == 1 ? 2 : 1)), DSIM_WIDE_WAVES(NW)) void dsim_bwd_kernel(KCommonT<O, D> k, const float* __restrict__ ckpt,
                                                           const float* __restrict__ act,
                                                           const float* __restrict__ mact,
                                                           const float* __restrict__ gq_out,
                                                           const float* __restrict__ gqd_out, float* gq_in,
                                                           float* gqd_in, float* gact, float* gmact, float* lit) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = dsim_env_index<MODE>();
    if (e >= k.n_envs) return;
    DevExec<NW, dsim_pf_regs<O, NW, LEAN, MODE>(), dsim_const_words<O>(), MODE == 1, MODE == 2 ? 2 : 1> ex;
    auto c = start_env<LEAN, MODE>(lds, k, k.o.total_words, ex);
    const size_t nq = k.d.nq, nd = k.d.nd, M = k.d.M;
    dsim_sim_step_backward(c, ex, k.substeps, k.mm_freq, ckpt + (size_t)e * k.ckpt_stride, act + e * nd,
                           M ? mact + e * M : nullptr, gq_out + e * nq, gqd_out + e * nd, gq_in + e * nq,
                           gqd_in + e * nd, gact ? gact + e * nd : nullptr, (gmact && M) ? gmact + e * M : nullptr,
                           lit ? lit + (size_t)e * (nq + nd + nd * nd) : nullptr);
}

// dsim_step_backward_literal, second launch: gq_in[quaternion block of joint j] += rho_j q_j (dsim_literal.hpp).  One thread per
// (environment, link); threads of links without a quaternion joint leave at once.  Run-time layout for every model.
// row_words: floats of one substep's checkpoint row in the model's checkpoint mode (both modes start a row with q, qd).
__global__ __launch_bounds__(64) void dsim_literal_radial_kernel(KCommonT<DsimOff, DsimDims> k, int row_words,
                                                                  const float* __restrict__ ckpt, const float* __restrict__ act,
                                                                  const float* __restrict__ mact, const float* __restrict__ lit,
                                                                  float* gq_in) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x), L = k.d.L;
    const int e = t / L, i = t - e * L;
    if (e >= k.n_envs) return;
    dsim_lit::Consts c;
    c.cb = k.cblob;
    c.o = k.o;
    c.d = k.d;
    const int type = c.I(k.o.jtype, i);
    if (type != DSIM_JOINT_BALL && type != DSIM_JOINT_FREE) return;
    const size_t nq = k.d.nq, nd = k.d.nd, M = k.d.M;
    const float* row = ckpt + (size_t)e * k.ckpt_stride;                  // the first substep's row: (q, qd) as handed in
    const float* hinv = row + (size_t)k.substeps * (size_t)row_words;      // the first group's H^-1
    const float* l = lit + (size_t)e * (nq + nd + nd * nd);
    // (the arrays of a row are padded to 16 bytes each: qd starts at its layout offset, not at nq)
    const float rho = dsim_lit::radial(c, row, row + (k.o.qd - k.o.q), act + e * nd, M ? mact + e * M : nullptr, k.h, hinv, l, l + nq, l + nq + nd, i);
    const int qs = c.I(k.o.qstart, i) + (type == DSIM_JOINT_FREE ? 3 : 0);
    float* g = gq_in + (size_t)e * nq + qs;
    for (int j = 0; j < 4; ++j) g[j] += rho * row[qs + j];
}

template <class O, class D, int NW, bool LEAN, int MODE>
__global__ __launch_bounds__((DSIM_NL * NW * (MODE == 1 ? 2 : 1)), (MODE == DSIM_MODE_PAIR ? DSIM_PAIR_WAVES : DSIM_WIDE_WAVES(NW))) void dsim_env_fwd_kernel(KCommonT<O, D> k, DsimEnvSpec sp, DsimEpisode ep,
                                                               const float* __restrict__ q_in,
                                                               const float* __restrict__ qd_in,
                                                               const float* __restrict__ actions, float* q_out,
                                                               float* qd_out, float* obs, float* rew, float* ckpt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = dsim_env_index<MODE>();
    if (e >= k.n_envs) return;
    DevExec<NW, dsim_pf_regs<O, NW, LEAN, MODE>(), dsim_const_words<O>(), MODE == 1, MODE == 2 ? 2 : 1> ex;
    auto c = start_env<LEAN, MODE>(lds, k, k.o.fwd_words, ex);
    const size_t nq = k.d.nq, nd = k.d.nd;
    dsim_env_fused_forward(c, ex, sp, k.substeps, k.mm_freq, q_in + e * nq, qd_in + e * nd, actions + (size_t)e * sp.n_act,
                           q_out + e * nq, qd_out + e * nd, obs + (size_t)e * sp.n_obs, rew + e,
                           ckpt ? ckpt + (size_t)e * k.ckpt_stride : nullptr, ep, e, k.n_envs, k.status);
}

template <class O, class D, int NW, bool LEAN, int MODE>
__global__ __launch_bounds__((DSIM_NL * NW * (MODE == 1 ? 2 : 1)), DSIM_WIDE_WAVES(NW)) void dsim_env_bwd_kernel(KCommonT<O, D> k, DsimEnvSpec sp,
                                                               const float* __restrict__ ckpt,
                                                               const float* __restrict__ actions,
                                                               const float* __restrict__ gq_out,
                                                               const float* __restrict__ gqd_out,
                                                               const float* __restrict__ gobs,
                                                               const float* __restrict__ grew,
                                                               const float* __restrict__ gobs_before, float* gq_in,
                                                               float* gqd_in, float* gactions) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = dsim_env_index<MODE>();
    if (e >= k.n_envs) return;
    DevExec<NW, dsim_pf_regs<O, NW, LEAN, MODE>(), dsim_const_words<O>(), MODE == 1, MODE == 2 ? 2 : 1> ex;
    auto c = start_env<LEAN, MODE>(lds, k, k.o.total_words, ex);
    const size_t nq = k.d.nq, nd = k.d.nd;
    dsim_env_fused_backward(c, ex, sp, k.substeps, k.mm_freq, ckpt + (size_t)e * k.ckpt_stride,
                            actions + (size_t)e * sp.n_act, gq_out ? gq_out + e * nq : nullptr,
                            gqd_out ? gqd_out + e * nd : nullptr, gobs ? gobs + (size_t)e * sp.n_obs : nullptr,
                            grew ? grew + e : nullptr, gobs_before ? gobs_before + (size_t)e * sp.n_obs : nullptr,
                            gq_in + e * nq, gqd_in + e * nd, gactions + (size_t)e * sp.n_act);
}

template <class O, class D, int NW>
__global__ __launch_bounds__(DSIM_NL * NW) void dsim_env_obs_kernel(KCommonT<O, D> k, DsimEnvSpec sp,
                                                               const float* __restrict__ q,
                                                               const float* __restrict__ qd,
                                                               const float* __restrict__ stored, float* obs,
                                                               float* rew) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = blockIdx.x;
    if (e >= k.n_envs) return;
    DsimCtxT<O, D> c;
    c.s = lds;
    c.k = lds;
    c.o = k.o;
    c.d = k.d;
    c.h = k.h;
    DevExec<NW> ex;
    dsim_env_observe_only(c, ex, sp, q + (size_t)e * k.d.nq, qd + (size_t)e * k.d.nd, stored + (size_t)e * sp.n_act,
                          obs + (size_t)e * sp.n_obs, rew + e);
}

template <class O, class D, int NW>
__global__ __launch_bounds__(DSIM_NL * NW) void dsim_body_xf_kernel(KCommonT<O, D> k, const float* __restrict__ q, float* xsc,
                                                                  float* xsm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = blockIdx.x;
    if (e >= k.n_envs) return;
    DevExec<NW, 6, dsim_const_words<O>(), false> ex;
    auto c = start_env<false, DSIM_MODE_PLAIN>(lds, k, k.o.fwd_words, ex);
    dsim_body_transforms_only(c, ex, q + (size_t)e * k.d.nq, xsc + (size_t)e * 7 * k.d.L, xsm ? xsm + (size_t)e * 7 * k.d.L : nullptr);
}

#ifdef DSIM_ENABLE_PHASE_TIMER
// developer tool (tools/phase_timer.py): per-phase cycle stamps of workgroup 0; NOT compiled into the product library
template <int NW> struct TimingExec {
    static constexpr int NL = DSIM_NL * NW;
    static constexpr bool WAVE_OPS = NW == 1;
    static constexpr bool WAVE0_OPS = true;
    template <class F> __device__ __forceinline__ void run_wave0(F&& f) {
        run([&](int lane) {
            if (lane < DSIM_NL) f(lane);
        });
    }
    __device__ __forceinline__ float shfl(float v, int src) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
    }
    __device__ __forceinline__ float bcast(float v, int src) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
    }
    __device__ __forceinline__ void lds_fence() { __syncthreads(); }
    __device__ __forceinline__ void system_fence() { __threadfence_system(); }
    long long* buf;
    int idx, cap;
    int tag = 0;
    static constexpr bool HAS_HELPER = false;
    template <class F> __device__ __forceinline__ void run_both(F&& f) { run(f); }
    template <class FM, class FH> __device__ __forceinline__ void fork_join(FM&& fm, FH&& fh) {
        run([&](int lane) {
            fm(lane);
            fh(lane);
        });
    }
    template <class FM, class FH> __device__ __forceinline__ void fork_join_mid(FM&& fm, FH&& fh) { fork_join(fm, fh); }
    template <class FM, class FH> __device__ __forceinline__ void fork_mid_detached(FM&& fm, FH&& fh) { fork_join(fm, fh); }
    template <class FM, class FH> __device__ __forceinline__ void fork_side(FM&& fm, FH&& fh) {
        run([&](int lane) {
            fh(lane);
            __syncthreads();
            fm(lane);
        });
    }
    __device__ __forceinline__ void side_done() { __syncthreads(); }
    template <int D> __device__ __forceinline__ float from_above(float v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + D, 0xf, 0xf, true));
    }
    template <int D> __device__ __forceinline__ float from_below(float v) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, true));
    }
    template <int D, bool FIRST = true>
    __device__ __forceinline__ void add_from_above(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        a0 = __builtin_fmaf(from_above<D>(a0), w, a0); a1 = __builtin_fmaf(from_above<D>(a1), w, a1);
        a2 = __builtin_fmaf(from_above<D>(a2), w, a2); a3 = __builtin_fmaf(from_above<D>(a3), w, a3);
        a4 = __builtin_fmaf(from_above<D>(a4), w, a4); a5 = __builtin_fmaf(from_above<D>(a5), w, a5);
    }
    template <bool FIRST = true>
    __device__ __forceinline__ void add_from_next(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        auto nx = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); };
        a0 = __builtin_fmaf(nx(a0), w, a0); a1 = __builtin_fmaf(nx(a1), w, a1); a2 = __builtin_fmaf(nx(a2), w, a2);
        a3 = __builtin_fmaf(nx(a3), w, a3); a4 = __builtin_fmaf(nx(a4), w, a4); a5 = __builtin_fmaf(nx(a5), w, a5);
    }
    template <int SRC>
    __device__ __forceinline__ void add_from_lane(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        const float s0 = bcast(a0, SRC), s1 = bcast(a1, SRC), s2 = bcast(a2, SRC), s3 = bcast(a3, SRC), s4 = bcast(a4, SRC), s5 = bcast(a5, SRC);
        a0 = __builtin_fmaf(s0, w, a0); a1 = __builtin_fmaf(s1, w, a1); a2 = __builtin_fmaf(s2, w, a2);
        a3 = __builtin_fmaf(s3, w, a3); a4 = __builtin_fmaf(s4, w, a4); a5 = __builtin_fmaf(s5, w, a5);
    }
    __device__ __forceinline__ void loads_landed() {}
    __device__ __forceinline__ void group_sync() {}
    __device__ __forceinline__ void helper_prefetch(const float*, int) {}
    __device__ __forceinline__ void helper_commit(float*, int, const float*) {}
    __device__ __forceinline__ void helper_prefetch_aux(const float*, int) {}
    __device__ __forceinline__ void helper_commit_aux(float*, float*, int) {}
    __device__ __forceinline__ void mid() { __syncthreads(); }
    __device__ __forceinline__ void mid2() { if constexpr (NW > 1) __syncthreads(); }
    __device__ __forceinline__ void side_done_w() { __syncthreads(); }
    template <class F0, class FR> __device__ __forceinline__ void fork_wave0(F0&& f0, FR&& fr) {
        run([&](int lane) {
            if (lane < DSIM_NL) f0(lane);
            else fr(lane - DSIM_NL);
        });
    }
    __device__ __forceinline__ void stamp() {}
    DsimImage<NW, 0> img_;
    __device__ __forceinline__ void begin_request() {}
    __device__ __forceinline__ void begin() { img_.land(); __syncthreads(); }
    float io_[DSIM_IO_MAX];
    __device__ __forceinline__ float* io(int) { return io_; }
    __device__ __forceinline__ void mark(int t) { tag = t * 100; }
    template <class F> __device__ __forceinline__ void run(F&& f) {
        f((int)threadIdx.x);
        __syncthreads();
        if (blockIdx.x == 0 && threadIdx.x == 0 && idx < cap) {
            buf[idx] = clock64();
            buf[cap + idx] = tag;
        }
        ++tag;
        ++idx;
    }
    template <class F> __device__ __forceinline__ void fire(F&& f) {
        f((int)threadIdx.x);
        if constexpr (NW > 1) __syncthreads();
    }
    template <int N> __device__ __forceinline__ void wave_gj(float* H) {
        run([&](int) { dsim_wave_gj<N, NW>(H); });
    }
    static constexpr bool WAVE_GJ_PAD = true;
    template <int NB> __device__ __forceinline__ void wave_gj_pad(float* H, int n) {
        run([&](int) { dsim_wave_gj_pad<NB, NW>(H, n); });
    }
    float hacc_[DSIM_HACC_MAX];
    __device__ __forceinline__ float* hacc(int) { return hacc_; }
    float hpf_[DSIM_HPF_MAX];
    __device__ __forceinline__ float* hpf(int) { return hpf_; }
    DsimTopoRegs topo_;
    __device__ __forceinline__ DsimTopoRegs& topo(int) { return topo_; }
    const float* pf_src;
    __device__ __forceinline__ void prefetch(const float* row, int) { pf_src = row; }
    __device__ __forceinline__ void commit(float* dst, int words, int lane) {
        for (int k = lane; k < words; k += NL) dst[k] = pf_src[k];
    }
    __device__ __forceinline__ void prefetch_rest(const float* row, int) { pf_src = row; }
    __device__ __forceinline__ void commit_rest(float* dst, int words, int lane) {
        for (int k = lane; k < words; k += NL - DSIM_NL) dst[k] = pf_src[k];
    }
};
template <class O, class D, int NW>
__global__ __launch_bounds__(DSIM_NL * NW) void dsim_timer_kernel(KCommonT<O, D> k, DsimEnvSpec sp, int backward,
                                                             const float* q_in, const float* qd_in, const float* actions,
                                                             float* q_out, float* qd_out, float* obs, float* rew,
                                                             float* ckpt, const float* gq_out, const float* gqd_out,
                                                             const float* gobs, const float* grew, float* gq_in,
                                                             float* gqd_in, float* gactions, long long* stamps, int cap) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int e = blockIdx.x;
    if (e >= k.n_envs) return;
    TimingExec<NW> ex{stamps, 1, cap};
    auto c = start_env<false, DSIM_MODE_PLAIN>(lds, k, k.o.total_words, ex);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = clock64();
    const size_t nq = k.d.nq, nd = k.d.nd;
    float* ck = ckpt + (size_t)e * k.ckpt_stride;
    if (!backward)
        dsim_env_fused_forward(c, ex, sp, k.substeps, k.mm_freq, q_in + e * nq, qd_in + e * nd,
                               actions + (size_t)e * sp.n_act, q_out + e * nq, qd_out + e * nd,
                               obs + (size_t)e * sp.n_obs, rew + e, ck, DsimEpisode{}, e, k.n_envs);
    else
        dsim_env_fused_backward(c, ex, sp, k.substeps, k.mm_freq, ck, actions + (size_t)e * sp.n_act, gq_out + e * nq,
                                gqd_out + e * nd, gobs + (size_t)e * sp.n_obs, grew + e, nullptr, gq_in + e * nq,
                                gqd_in + e * nd, gactions + (size_t)e * sp.n_act);
}
#endif

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return DSIM_ERR_HIP;
}

// ---- kernel variants -------------------------------------------------------------------------------
enum Variant {
    V_GENERIC = 0,
#define DSIM_ENUM(T) V_##T,
    DSIM_STATIC_VARIANTS(DSIM_ENUM)
#undef DSIM_ENUM
};

int match_variant(const DsimLayout& lay) {
    if (getenv("DSIM_FORCE_GENERIC")) return V_GENERIC;
    const size_t no = sizeof(DsimOff) / sizeof(int), nd = sizeof(DsimDims) / sizeof(int);
#define DSIM_MATCH(T)                                                                                              \
    if (sizeof(kDsimStatic##T) == (no + nd) * sizeof(int) && memcmp(kDsimStatic##T, &lay.o, no * sizeof(int)) == 0 && \
        memcmp(kDsimStatic##T + no, &lay.d, nd * sizeof(int)) == 0)                                                 \
        return V_##T;
    DSIM_STATIC_VARIANTS(DSIM_MATCH)
#undef DSIM_MATCH
    return V_GENERIC;
}

}  // namespace

struct dsim_model {
    DsimLayout lay;
    uint32_t* d_cblob = nullptr;
    int variant = V_GENERIC;
    int waves = 1;   // wavefronts per environment: 1 or DSIM_WAVES_WIDE
    int lean = 0;    // checkpoint mode (dsim_model_set_ckpt_mode)
    int device = 0;  // HIP device the constants live on (the current device of dsim_model_create); every call checks it
    // Status words written by the kernels (dsim_core.hpp: dsim_check_unit_quats): [0] != 0 -- a launch was handed a non-unit
    // quaternion, [1] -- the environment that saw it.  Pinned host memory mapped into the device: a kernel's store lands here
    // without any copy, and the next call on the model reads it without touching the stream (no synchronisation anywhere).
    volatile int* h_status = nullptr;
    int* d_status = nullptr;
    int row_words() const { return lean ? lay.o.xsc - lay.o.q : lay.o.save_words; }
    // Helper-wave kernels (DevExec<..., HELPER>) are used while every environment of the launch is resident at once: the
    // helper takes a wave slot of its SIMD, so beyond that point (several rounds of workgroups) it would halve the
    // number of environments in flight -- measured: Ant 8192 envs 13.4 M env-steps/s without, 8.1 M with.
    int helper_max_envs = 0;
    bool helper_ok(int n_envs) const { return n_envs <= helper_max_envs; }
    // Two environments per wavefront (DSIM_MODE_PAIR; forward kernels of models with pair kernels only -- dsim_with_flags falls
    // back to the plain kernels for the others): every launch beyond the helper-wave capacity.  A pair wave is a few per cent
    // longer than a plain one (items beyond 32 lanes take two passes) and carries two environments: half the waves per launch,
    // and a third more environments resident per CU (Ant: 16 instead of 12).  DSIM_PAIR=0 / 1 forces the choice (A/B runs).
    int pair_min_envs = 1 << 30;
    int mode(int n_envs, bool fwd) const {
        if (helper_ok(n_envs)) return DSIM_MODE_HELPER;
#ifndef DSIM_PAIR_BWD
        if (!fwd) return DSIM_MODE_PLAIN;
#endif
        return n_envs >= pair_min_envs ? DSIM_MODE_PAIR : DSIM_MODE_PLAIN;
    }
};
void dsim_helper_capacity(dsim_model* m);
#ifndef DSIM_WAVES_WIDE
#define DSIM_WAVES_WIDE 4   // (-DDSIM_WAVES_WIDE=2 builds the A/B variant)
#endif

namespace {

// calls f(offsets, dims, waves) with the (static or runtime) layout types of the model's kernel variant and the number of
// wavefronts per environment as a compile-time constant
template <class D> constexpr bool dsim_static_wide() { return D::NS > DSIM_NL || D::C > DSIM_NL; }
template <class F> int dispatch(const dsim_model* m, F&& f) {
    const bool wide = m->waves > 1;
    switch (m->variant) {
    // a specialised variant exists with the wave count its model wants (pick_waves), not with both
#define DSIM_CASE(T)                                                                                        \
    case V_##T:                                                                                             \
        if constexpr (dsim_static_wide<DsimDims##T>())                                                      \
            return f(DsimOff##T{}, DsimDims##T{}, std::integral_constant<int, DSIM_WAVES_WIDE>{});          \
        else                                                                                                \
            return f(DsimOff##T{}, DsimDims##T{}, std::integral_constant<int, 1>{});
        DSIM_STATIC_VARIANTS(DSIM_CASE)
#undef DSIM_CASE
        default:
            return wide ? f(m->lay.o, m->lay.d, std::integral_constant<int, DSIM_WAVES_WIDE>{})
                        : f(m->lay.o, m->lay.d, std::integral_constant<int, 1>{});
    }
}

// Wavefronts per environment.  One wave is the lowest-latency mapping while the per-item phases are one or two passes
// over a wavefront (Ant, Humanoid, the planar models, cartpole): a workgroup barrier costs more than a second pass, and
// 4 waves per environment need 4 resident waves per SIMD at 1024 environments, which the kernels' ~200 registers per
// lane do not allow (measured: Humanoid forward 0.45 -> 0.90 ms).  Models with item lists several wavefronts long
// (SNUHumanoid: 198 muscle segments, 88 contacts) run 4 waves per environment: their long phases become single
// passes on the CU's 4 SIMDs (SNUHumanoid 512 envs: 0.62 + 0.98 -> 0.45 + 0.62 ms).  DSIM_WAVES=1|4 overrides the
// choice for the generic kernels (A/B runs: DSIM_FORCE_GENERIC=1 DSIM_WAVES=...).
int pick_waves(const DsimLayout& lay, int variant) {
    const DsimDims& d = lay.d;
    const int want = (d.NS > DSIM_NL || d.C > DSIM_NL) ? DSIM_WAVES_WIDE : 1;
    if (variant != V_GENERIC) return want;   // specialised kernels are compiled for their model's wave count only
    if (const char* e = getenv("DSIM_WAVES")) return atoi(e) > 1 ? DSIM_WAVES_WIDE : 1;
    return want;
}

template <class O, class D>
KCommonT<O, D> make_k(const dsim_model* m, O o, D d, int n_envs, float dt, int substeps, int mm_freq) {
    KCommonT<O, D> k;
    k.o = o;
    k.d = d;
    k.cblob = m->d_cblob;
    k.h = dt / float(substeps);
    k.substeps = substeps;
    k.mm_freq = mm_freq;
    k.n_envs = n_envs;
    k.ckpt_stride = dsim_ckpt_words(m->row_words(), m->lay.d.nq, m->lay.d.nd, substeps, mm_freq);
    k.status = m->d_status;
    return k;
}

// a model belongs to the device it was created on: its constant block lives there and the kernel attributes / occupancy
// figures were set up for it.  One process may drive several GPUs (one model per GPU); a call must run with its model's
// device current, and the stream and pointers it is given must belong to that device.
int check_device(const dsim_model* m) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev != m->device)
        return fail(DSIM_ERR_INVALID, "model was created on HIP device " + std::to_string(m->device) + " but the current device is " +
                                          std::to_string(dev) + " (hipSetDevice / torch.cuda.device before the call)");
    return DSIM_OK;
}

// A launch that was handed a non-unit quaternion leaves a mark in the model's status words; the next call that would launch
// on the model reports it (once: the mark is cleared) instead of launching.  Asynchronous, like a HIP error of a kernel:
// the state the marked launch returned is what the formulas give off the manifold, not what the reference gives there.
int check_status(const dsim_model* m) {
    if (m->h_status && m->h_status[0] != 0) {
        const int env = m->h_status[1];
        m->h_status[0] = 0;
        return fail(DSIM_ERR_INVALID, "an earlier launch on this model was given a joint_q whose quaternion block is not a unit "
                                      "quaternion (| |q| - 1 | > 1e-4; first seen in environment " + std::to_string(env) +
                                          "): the path is defined on unit quaternions only (include/dsim.h, dsim_step_forward); "
                                          "normalise the state before handing it in");
    }
    return DSIM_OK;
}

int check_common(const dsim_model* m, int n_envs, float dt, int substeps, int mm_freq) {
    if (!m) return fail(DSIM_ERR_INVALID, "null model");
    if (int rc = check_device(m)) return rc;
    if (int rc = check_status(m)) return rc;
    if (n_envs <= 0) return fail(DSIM_ERR_INVALID, "n_envs must be positive");
    if (substeps <= 0 || mm_freq <= 0) return fail(DSIM_ERR_INVALID, "substeps and mm_freq must be positive");
    if (!(dt > 0.f)) return fail(DSIM_ERR_INVALID, "dt must be positive");
    return DSIM_OK;
}

int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return DSIM_OK;
}

int make_spec(const dsim_model* m, const dsim_env_spec* e, DsimEnvSpec& sp) {
    if (!e) return fail(DSIM_ERR_INVALID, "null env spec");
    const DsimDims& d = m->lay.d;
    if (e->kind != DSIM_ENV_LOCOMOTION && e->kind != DSIM_ENV_CARTPOLE && e->kind != DSIM_ENV_PLANAR)
        return fail(DSIM_ERR_INVALID, "unknown env kind");
    if (e->n_act <= 0 || e->n_act > (d.M > d.nd ? d.M : d.nd)) return fail(DSIM_ERR_INVALID, "n_act out of range");
    if (!e->act_scale) return fail(DSIM_ERR_INVALID, "null act_scale");
    if (e->act_muscle) {
        if (e->n_act != d.M) return fail(DSIM_ERR_INVALID, "muscle actions must match the muscle count");
    } else if (e->act_offset < 0 || e->act_offset + e->n_act > d.nd) {
        return fail(DSIM_ERR_INVALID, "action dofs out of range");
    }
    int expect;
    if (e->kind == DSIM_ENV_LOCOMOTION) {
        if (d.nq < 7 || d.nd < 6 || m->lay.cblob[m->lay.o.jtype] != DSIM_JOINT_FREE)
            return fail(DSIM_ERR_INVALID, "locomotion observations need a free-floating root joint");
        expect = 13 + (d.nq - 7) + (d.nd - 6) + (e->obs_actions ? e->n_act : 0);
    } else if (e->kind == DSIM_ENV_PLANAR) {
        if (d.nq != d.nd || d.nq < 3) return fail(DSIM_ERR_INVALID, "planar observations need a 3-dof planar root");
        expect = d.nq - 1 + d.nd;
    } else {
        if (d.nq != 2 || d.nd != 2) return fail(DSIM_ERR_INVALID, "cartpole observations need 2 coordinates");
        expect = 5;
    }
    if (e->n_obs != expect) return fail(DSIM_ERR_INVALID, "n_obs does not match the observation layout");
    sp.kind = e->kind; sp.rew_kind = e->rew_kind; sp.n_act = e->n_act; sp.n_obs = e->n_obs;
    sp.act_offset = e->act_offset; sp.act_muscle = e->act_muscle; sp.obs_actions = e->obs_actions; sp.sanitize = e->sanitize_grads;
    for (int k = 0; k < 4; ++k) { sp.isr[k] = e->inv_start_rot[k]; sp.pen[k] = e->cartpole_penalties[k]; }
    sp.tgt_x = e->target_x; sp.tgt_z = e->target_z; sp.term_h = e->termination_height;
    sp.term_tol = e->termination_tolerance; sp.h_scale = e->height_rew_scale; sp.act_pen = e->action_penalty;
    sp.vel_scale = e->joint_vel_obs_scaling; sp.act_scale = e->act_scale;
    return DSIM_OK;
}

}  // namespace

extern "C" {

const char* dsim_last_error(void) { return g_err.c_str(); }
int dsim_version(void) { return 107; }

int dsim_model_create(const dsim_model_desc* desc, dsim_model** out) {
    if (!desc || !out) return fail(DSIM_ERR_INVALID, "null argument");
    dsim_model* m = new (std::nothrow) dsim_model();
    if (!m) return fail(DSIM_ERR_INVALID, "out of host memory");
    std::string err = dsim_build_layout(*desc, m->lay);
    if (!err.empty()) {
        delete m;
        return fail(DSIM_ERR_INVALID, err);
    }
    const int bytes = m->lay.o.total_words * 4, fbytes = m->lay.o.fwd_words * 4;
    if (bytes > 160 * 1024) {
        delete m;
        return fail(DSIM_ERR_LIMIT, "model needs more than 160 KiB of LDS per environment");
    }
    m->variant = match_variant(m->lay);
    m->waves = pick_waves(m->lay, m->variant);
    hipError_t e = hipGetDevice(&m->device);
    if (e == hipSuccess) {
        void* hp = nullptr;
        e = hipHostMalloc(&hp, 2 * sizeof(int), hipHostMallocMapped);
        if (e == hipSuccess) {
            m->h_status = static_cast<volatile int*>(hp);
            m->h_status[0] = m->h_status[1] = 0;
            void* dp = nullptr;
            e = hipHostGetDevicePointer(&dp, hp, 0);
            m->d_status = static_cast<int*>(dp);
        }
    }
    if (e == hipSuccess) e = hipMalloc(&m->d_cblob, sizeof(uint32_t) * m->lay.cblob.size());
    if (e == hipSuccess)
        e = hipMemcpy(m->d_cblob, m->lay.cblob.data(), sizeof(uint32_t) * m->lay.cblob.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess && (2 * bytes > 64 * 1024 || 2 * fbytes > 64 * 1024)) {   // (2 x: pair launches)
        // Opt in to > 64 KiB of dynamic LDS.  The attribute belongs to the kernel FUNCTION, not to this model: two models
        // that share a kernel variant (e.g. two user models on the generic kernels) must not lower each other's limit,
        // so it is set once to the hardware maximum (160 KiB); what a launch actually gets is its own byte count.
        dispatch(m, [&](auto o, auto d, auto nw) {
            using O = decltype(o);
            using D = decltype(d);
            constexpr int NW = decltype(nw)::value;
            auto raise = [&](auto mode_c) {
                constexpr int HELP = decltype(mode_c)::value;
                const void* fns[] = {reinterpret_cast<const void*>(dsim_bwd_kernel<O, D, NW, false, HELP>),
                                     reinterpret_cast<const void*>(dsim_env_bwd_kernel<O, D, NW, false, HELP>),
                                     reinterpret_cast<const void*>(dsim_fwd_kernel<O, D, NW, false, HELP>),
                                     reinterpret_cast<const void*>(dsim_env_fwd_kernel<O, D, NW, false, HELP>),
                                     reinterpret_cast<const void*>(dsim_bwd_kernel<O, D, NW, true, HELP>),
                                     reinterpret_cast<const void*>(dsim_env_bwd_kernel<O, D, NW, true, HELP>),
                                     reinterpret_cast<const void*>(dsim_fwd_kernel<O, D, NW, true, HELP>),
                                     reinterpret_cast<const void*>(dsim_env_fwd_kernel<O, D, NW, true, HELP>)};
                for (const void* fn : fns)
                    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            };
            raise(std::integral_constant<int, DSIM_MODE_PLAIN>{});
            if constexpr (dsim_has_helper<D, NW, true>()) raise(std::integral_constant<int, DSIM_MODE_HELPER>{});
            else if constexpr (dsim_has_helper<D, NW, false>()) {   // (generic kernels: helper-wave ADJOINT kernels only)
                const void* fns[] = {reinterpret_cast<const void*>(dsim_bwd_kernel<O, D, NW, false, DSIM_MODE_HELPER>),
                                     reinterpret_cast<const void*>(dsim_env_bwd_kernel<O, D, NW, false, DSIM_MODE_HELPER>),
                                     reinterpret_cast<const void*>(dsim_bwd_kernel<O, D, NW, true, DSIM_MODE_HELPER>),
                                     reinterpret_cast<const void*>(dsim_env_bwd_kernel<O, D, NW, true, DSIM_MODE_HELPER>)};
                for (const void* fn : fns)
                    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            if constexpr (dsim_has_pair<D, NW>()) {
                const void* fns[] = {reinterpret_cast<const void*>(dsim_fwd_kernel<O, D, NW, false, DSIM_MODE_PAIR>),
                                     reinterpret_cast<const void*>(dsim_env_fwd_kernel<O, D, NW, false, DSIM_MODE_PAIR>),
                                     reinterpret_cast<const void*>(dsim_fwd_kernel<O, D, NW, true, DSIM_MODE_PAIR>),
                                     reinterpret_cast<const void*>(dsim_env_fwd_kernel<O, D, NW, true, DSIM_MODE_PAIR>)};
                for (const void* fn : fns)
                    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            }
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(dsim_env_obs_kernel<O, D, NW>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(dsim_body_xf_kernel<O, D, NW>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            return 0;
        });
    }
    if (e != hipSuccess) {
        if (m->d_cblob) (void)hipFree(m->d_cblob);
        if (m->h_status) (void)hipHostFree(const_cast<int*>(m->h_status));
        delete m;
        return hip_fail(e, "dsim_model_create");
    }
    dsim_helper_capacity(m);
    *out = m;
    return DSIM_OK;
}

}  // extern "C"

// helper-wave kernels: how many environments are resident at once (the smallest figure over the model's kernels in the
// current checkpoint mode); DSIM_HELPER=0 / 1 forces the choice (A/B runs)
void dsim_helper_capacity(dsim_model* m) {
    dispatch(m, [&](auto o, auto d, auto nw) {
        using O = decltype(o);
        using D = decltype(d);
        constexpr int NW = decltype(nw)::value;
        if constexpr (dsim_has_helper<D, NW>()) {
            int cus = 0, per_cu = 1 << 20;   // (callers have the model's device current: dsim_model_create, check_device)
            bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device) == hipSuccess;
            auto cap = [&](auto kernel, int words) {   // resident workgroups per CU of one helper kernel
                int n = 0;
                ok = ok && hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 2 * DSIM_NL, (size_t)words * 4) == hipSuccess;
                if (n < per_cu) per_cu = n;
            };
            constexpr bool FWD_TOO = dsim_has_helper<D, NW, true>();   // (generic kernels: the adjoint alone)
            cap(dsim_env_bwd_kernel<O, D, NW, false, DSIM_MODE_HELPER>, m->lay.o.total_words);
            cap(dsim_bwd_kernel<O, D, NW, false, DSIM_MODE_HELPER>, m->lay.o.total_words);
            if constexpr (FWD_TOO) {
                cap(dsim_env_fwd_kernel<O, D, NW, false, DSIM_MODE_HELPER>, m->lay.o.fwd_words);
                cap(dsim_fwd_kernel<O, D, NW, false, DSIM_MODE_HELPER>, m->lay.o.fwd_words);
            }
            if (m->lean) {   // (the mode is chosen right after creation; dsim_model_set_ckpt_mode re-evaluates)
                cap(dsim_env_bwd_kernel<O, D, NW, true, DSIM_MODE_HELPER>, m->lay.o.total_words);
                cap(dsim_bwd_kernel<O, D, NW, true, DSIM_MODE_HELPER>, m->lay.o.total_words);
                if constexpr (FWD_TOO) {
                    cap(dsim_env_fwd_kernel<O, D, NW, true, DSIM_MODE_HELPER>, m->lay.o.fwd_words);
                    cap(dsim_fwd_kernel<O, D, NW, true, DSIM_MODE_HELPER>, m->lay.o.fwd_words);
                }
            }
            m->helper_max_envs = ok ? cus * per_cu : 0;
            if (const char* f = getenv("DSIM_HELPER")) m->helper_max_envs = atoi(f) ? (1 << 30) : 0;
        }
        if constexpr (dsim_has_pair<D, NW>()) {
            m->pair_min_envs = 0;
            if (const char* f = getenv("DSIM_PAIR")) m->pair_min_envs = atoi(f) ? 0 : (1 << 30);
        }
        return 0;
    });
}

extern "C" {

int dsim_model_destroy(dsim_model* m) {
    if (!m) return DSIM_OK;
    if (m->d_cblob) (void)hipFree(m->d_cblob);
    if (m->h_status) (void)hipHostFree(const_cast<int*>(m->h_status));
    delete m;
    return DSIM_OK;
}

int dsim_model_status(dsim_model* m, int* first_env) {
    if (!m) return fail(DSIM_ERR_INVALID, "null model");
    if (first_env) *first_env = -1;
    if (m->h_status && m->h_status[0] != 0) {
        if (first_env) *first_env = m->h_status[1];
        return check_status(m);
    }
    return DSIM_OK;
}

int64_t dsim_ckpt_floats_mm(const dsim_model* m, int substeps, int mm_freq) {
    if (!m || substeps <= 0 || mm_freq <= 0) return 0;
    return (int64_t)dsim_ckpt_words(m->row_words(), m->lay.d.nq, m->lay.d.nd, substeps, mm_freq);
}

int64_t dsim_ckpt_floats(const dsim_model* m, int substeps) { return dsim_ckpt_floats_mm(m, substeps, 1); }

int dsim_model_set_ckpt_mode(dsim_model* m, int mode) {
    if (!m) return fail(DSIM_ERR_INVALID, "null model");
    if (mode != DSIM_CKPT_FULL && mode != DSIM_CKPT_LEAN) return fail(DSIM_ERR_INVALID, "unknown checkpoint mode");
    if (int rc = check_device(m)) return rc;   // the occupancy query below is per device
    m->lean = mode == DSIM_CKPT_LEAN;
    dsim_helper_capacity(m);
    return DSIM_OK;
}

int dsim_step_forward(const dsim_model* m, int n_envs, const float* q_in, const float* qd_in, const float* act,
                      const float* muscle_act, float dt, int substeps, int mm_freq, float* q_out, float* qd_out,
                      float* ckpt, void* hip_stream) {
    int rc = check_common(m, n_envs, dt, substeps, mm_freq);
    if (rc) return rc;
    if (!q_in || !qd_in || !act || !q_out || !qd_out) return fail(DSIM_ERR_INVALID, "null state pointer");
    if (m->lay.d.M > 0 && !muscle_act) return fail(DSIM_ERR_INVALID, "model has muscles but muscle_act is null");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, dt, substeps, mm_freq);
        dsim_with_flags<decltype(d), NW, true>(m->lean, m->mode(n_envs, true), [&](auto lean_c, auto mode_c) {
            constexpr bool LEAN = decltype(lean_c)::value;
            constexpr int MODE = decltype(mode_c)::value, EPB = MODE == DSIM_MODE_PAIR ? 2 : 1;
            hipLaunchKernelGGL((dsim_fwd_kernel<decltype(o), decltype(d), NW, LEAN, MODE>), dim3((n_envs + EPB - 1) / EPB),
                           dim3(DSIM_NL * NW * (MODE == DSIM_MODE_HELPER ? 2 : 1)), ((size_t)m->lay.o.fwd_words * EPB - (size_t)m->lay.o.const_words * (EPB - 1)) * 4, st, k, q_in, qd_in, act, muscle_act, q_out, qd_out, ckpt);
            return 0;
        });
        return launched("launch dsim_fwd_kernel");
    });
}

static int step_backward(const dsim_model* m, int n_envs, const float* ckpt, const float* act, const float* muscle_act,
                         float dt, int substeps, int mm_freq, const float* gq_out, const float* gqd_out, float* gq_in,
                         float* gqd_in, float* gact, float* gmuscle_act, float* lit, void* hip_stream) {
    int rc = check_common(m, n_envs, dt, substeps, mm_freq);
    if (rc) return rc;
    if (!ckpt || !act || !gq_out || !gqd_out || !gq_in || !gqd_in)
        return fail(DSIM_ERR_INVALID, "null pointer (ckpt/act/grad)");
    if (m->lay.d.M > 0 && !muscle_act) return fail(DSIM_ERR_INVALID, "model has muscles but muscle_act is null");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, dt, substeps, mm_freq);
        dsim_with_flags<decltype(d), NW, false>(m->lean, m->mode(n_envs, false), [&](auto lean_c, auto mode_c) {
            constexpr bool LEAN = decltype(lean_c)::value;
            constexpr int MODE = decltype(mode_c)::value, EPB = MODE == DSIM_MODE_PAIR ? 2 : 1;
            hipLaunchKernelGGL((dsim_bwd_kernel<decltype(o), decltype(d), NW, LEAN, MODE>), dim3((n_envs + EPB - 1) / EPB),
                           dim3(DSIM_NL * NW * (MODE == DSIM_MODE_HELPER ? 2 : 1)), ((size_t)m->lay.o.total_words * EPB - (size_t)m->lay.o.const_words * (EPB - 1)) * 4, st, k, ckpt, act, muscle_act, gq_out, gqd_out, gq_in, gqd_in,
                           gact, gmuscle_act, lit);
            return 0;
        });
        return launched("launch dsim_bwd_kernel");
    });
}

int dsim_step_backward(const dsim_model* m, int n_envs, const float* ckpt, const float* act, const float* muscle_act,
                       float dt, int substeps, int mm_freq, const float* gq_out, const float* gqd_out, float* gq_in,
                       float* gqd_in, float* gact, float* gmuscle_act, void* hip_stream) {
    return step_backward(m, n_envs, ckpt, act, muscle_act, dt, substeps, mm_freq, gq_out, gqd_out, gq_in, gqd_in, gact, gmuscle_act,
                         nullptr, hip_stream);
}

int64_t dsim_literal_scratch_floats(const dsim_model* m) {
    if (!m) return 0;
    const int64_t nq = m->lay.d.nq, nd = m->lay.d.nd;
    return nq + nd + nd * nd;
}

int dsim_step_backward_literal(const dsim_model* m, int n_envs, const float* ckpt, const float* act, const float* muscle_act,
                               float dt, int substeps, int mm_freq, const float* gq_out, const float* gqd_out, float* gq_in,
                               float* gqd_in, float* gact, float* gmuscle_act, float* scratch, void* hip_stream) {
    if (!m) return fail(DSIM_ERR_INVALID, "null model");
    if (!scratch) return fail(DSIM_ERR_INVALID, "null scratch (n_envs x dsim_literal_scratch_floats(m) floats)");
    if (m->lay.d.L > DSIM_LIT_LMAX || m->lay.d.nd > DSIM_LIT_NDMAX)
        return fail(DSIM_ERR_LIMIT, "dsim_step_backward_literal handles models of up to " + std::to_string(DSIM_LIT_LMAX) + " links and " +
                                        std::to_string(DSIM_LIT_NDMAX) + " dofs");
    int rc = step_backward(m, n_envs, ckpt, act, muscle_act, dt, substeps, mm_freq, gq_out, gqd_out, gq_in, gqd_in, gact, gmuscle_act,
                           scratch, hip_stream);
    if (rc) return rc;
    if ((m->lay.d.tmask & (DSIM_TM(DSIM_JOINT_BALL) | DSIM_TM(DSIM_JOINT_FREE))) == 0) return DSIM_OK;   // no quaternion coordinates
    auto k = make_k(m, m->lay.o, m->lay.d, n_envs, dt, substeps, mm_freq);
    const long long threads = (long long)n_envs * m->lay.d.L;
    hipLaunchKernelGGL(dsim_literal_radial_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(hip_stream), k,
                       m->row_words(), ckpt, act, muscle_act, scratch, gq_in);
    return launched("launch dsim_literal_radial_kernel");
}

int dsim_env_step_forward(const dsim_model* m, const dsim_env_spec* env, int n_envs, const float* q_in,
                          const float* qd_in, const float* actions, float dt, int substeps, int mm_freq, float* q_out,
                          float* qd_out, float* obs, float* rew, float* ckpt, const dsim_episode* episode,
                          void* hip_stream) {
    int rc = check_common(m, n_envs, dt, substeps, mm_freq);
    if (rc) return rc;
    DsimEnvSpec sp;
    rc = make_spec(m, env, sp);
    if (rc) return rc;
    if (!q_in || !qd_in || !actions || !q_out || !qd_out || !obs || !rew) return fail(DSIM_ERR_INVALID, "null pointer");
    DsimEpisode ep{};
    if (episode) {
        if (!episode->progress || !episode->done || !episode->reset_q || !episode->reset_qd || !episode->reset_count)
            return fail(DSIM_ERR_INVALID, "episode: null pointer (progress/done/reset_q/reset_qd/reset_count)");
        if (episode->reset_pool <= 0) return fail(DSIM_ERR_INVALID, "episode: reset_pool must be positive");
        if (episode->episode_length <= 0) return fail(DSIM_ERR_INVALID, "episode: episode_length must be positive");
        ep.progress = reinterpret_cast<long long*>(episode->progress);
        ep.done = reinterpret_cast<long long*>(episode->done);
        ep.obs_before = episode->obs_before_reset;
        ep.reset_q = episode->reset_q;
        ep.reset_qd = episode->reset_qd;
        ep.reset_count = episode->reset_count;
        ep.pool = episode->reset_pool;
        ep.episode_length = episode->episode_length;
        ep.height_terminate = episode->height_terminate;
        ep.check_invalid = episode->check_invalid;
        ep.noise_q = episode->noise_q;
        ep.noise_qd = episode->noise_qd;
        ep.noise_angle = episode->noise_angle;
        ep.seed = episode->seed;
    }
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, dt, substeps, mm_freq);
        dsim_with_flags<decltype(d), NW, true>(m->lean, m->mode(n_envs, true), [&](auto lean_c, auto mode_c) {
            constexpr bool LEAN = decltype(lean_c)::value;
            constexpr int MODE = decltype(mode_c)::value, EPB = MODE == DSIM_MODE_PAIR ? 2 : 1;
            hipLaunchKernelGGL((dsim_env_fwd_kernel<decltype(o), decltype(d), NW, LEAN, MODE>), dim3((n_envs + EPB - 1) / EPB),
                           dim3(DSIM_NL * NW * (MODE == DSIM_MODE_HELPER ? 2 : 1)), ((size_t)m->lay.o.fwd_words * EPB - (size_t)m->lay.o.const_words * (EPB - 1)) * 4, st, k, sp, ep, q_in, qd_in, actions, q_out, qd_out, obs, rew,
                           ckpt);
            return 0;
        });
        return launched("launch dsim_env_fwd_kernel");
    });
}

int dsim_env_step_backward(const dsim_model* m, const dsim_env_spec* env, int n_envs, const float* ckpt,
                           const float* actions, float dt, int substeps, int mm_freq, const float* gq_out,
                           const float* gqd_out, const float* gobs, const float* grew, const float* gobs_before_reset,
                           float* gq_in, float* gqd_in, float* gactions, void* hip_stream) {
    int rc = check_common(m, n_envs, dt, substeps, mm_freq);
    if (rc) return rc;
    DsimEnvSpec sp;
    rc = make_spec(m, env, sp);
    if (rc) return rc;
    if (!ckpt || !actions || !gq_in || !gqd_in || !gactions) return fail(DSIM_ERR_INVALID, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, dt, substeps, mm_freq);
        dsim_with_flags<decltype(d), NW, false>(m->lean, m->mode(n_envs, false), [&](auto lean_c, auto mode_c) {
            constexpr bool LEAN = decltype(lean_c)::value;
            constexpr int MODE = decltype(mode_c)::value, EPB = MODE == DSIM_MODE_PAIR ? 2 : 1;
            hipLaunchKernelGGL((dsim_env_bwd_kernel<decltype(o), decltype(d), NW, LEAN, MODE>), dim3((n_envs + EPB - 1) / EPB),
                           dim3(DSIM_NL * NW * (MODE == DSIM_MODE_HELPER ? 2 : 1)), ((size_t)m->lay.o.total_words * EPB - (size_t)m->lay.o.const_words * (EPB - 1)) * 4, st, k, sp, ckpt, actions, gq_out, gqd_out, gobs, grew,
                           gobs_before_reset, gq_in, gqd_in, gactions);
            return 0;
        });
        return launched("launch dsim_env_bwd_kernel");
    });
}

int dsim_env_observe(const dsim_model* m, const dsim_env_spec* env, int n_envs, const float* q, const float* qd,
                     const float* stored_actions, float* obs, float* rew, void* hip_stream) {
    int rc = check_common(m, n_envs, 1.0f, 1, 1);
    if (rc) return rc;
    DsimEnvSpec sp;
    rc = make_spec(m, env, sp);
    if (rc) return rc;
    if (!q || !qd || !stored_actions || !obs || !rew) return fail(DSIM_ERR_INVALID, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, 1.0f, 1, 1);
        hipLaunchKernelGGL((dsim_env_obs_kernel<decltype(o), decltype(d), NW>), dim3(n_envs), dim3(DSIM_NL * NW),
                           (size_t)m->lay.o.fwd_words * 4, st, k, sp, q, qd, stored_actions, obs, rew);
        return launched("launch dsim_env_obs_kernel");
    });
}

int dsim_body_transforms(const dsim_model* m, int n_envs, const float* q, float* X_sc, float* X_sm, void* hip_stream) {
    int rc = check_common(m, n_envs, 1.0f, 1, 1);
    if (rc) return rc;
    if (!q || !X_sc) return fail(DSIM_ERR_INVALID, "null pointer");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, 1.0f, 1, 1);
        k.status = nullptr;   // a read-back of whatever state the caller holds: no precondition to enforce here
        hipLaunchKernelGGL((dsim_body_xf_kernel<decltype(o), decltype(d), NW>), dim3(n_envs), dim3(DSIM_NL * NW),
                           (size_t)m->lay.o.fwd_words * 4, st, k, q, X_sc, X_sm);
        return launched("launch dsim_body_xf_kernel");
    });
}

/* 0 = generic kernels, >0 = index of the specialised variant in use (diagnostics / tests) */
int dsim_model_variant(const dsim_model* m) { return m ? m->variant : -1; }
int dsim_model_device(const dsim_model* m) { return m ? m->device : -1; }

#ifdef DSIM_STAMPS
int dsim_debug_stamps(long long* out, int n) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dsim_stamps), sizeof(long long) * (size_t)(n < 2 * 16384 ? n : 2 * 16384));
    if (e == hipSuccess) {   // (cleared for the next launch: a shorter launch would otherwise end in the previous one's stamps)
        void* sym = nullptr;
        e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_dsim_stamps));
        if (e == hipSuccess) e = hipMemset(sym, 0, sizeof(long long) * 2 * 16384);
    }
    return e == hipSuccess ? DSIM_OK : hip_fail(e, "dsim_debug_stamps");
}
#endif
#ifdef DSIM_ENABLE_PHASE_TIMER
int dsim_debug_phase_timer(const dsim_model* m, const dsim_env_spec* env, int n_envs, int backward, const float* q_in,
                           const float* qd_in, const float* actions, float dt, int substeps, int mm_freq, float* q_out,
                           float* qd_out, float* obs, float* rew, float* ckpt, const float* gq_out, const float* gqd_out,
                           const float* gobs, const float* grew, float* gq_in, float* gqd_in, float* gactions,
                           long long* stamps, int cap, void* hip_stream) {
    int rc = check_common(m, n_envs, dt, substeps, mm_freq);
    if (rc) return rc;
    DsimEnvSpec sp;
    rc = make_spec(m, env, sp);
    if (rc) return rc;
    return dispatch(m, [&](auto o, auto d, auto nw) {
        constexpr int NW = decltype(nw)::value;
        auto k = make_k(m, o, d, n_envs, dt, substeps, mm_freq);
        hipLaunchKernelGGL((dsim_timer_kernel<decltype(o), decltype(d), NW>), dim3(n_envs), dim3(DSIM_NL * NW),
                           (size_t)m->lay.o.total_words * 4, static_cast<hipStream_t>(hip_stream), k, sp, backward, q_in,
                           qd_in, actions, q_out, qd_out, obs, rew, ckpt, gq_out, gqd_out, gobs, grew, gq_in, gqd_in, gactions,
                           stamps, cap);
        return launched("launch dsim_timer_kernel");
    });
}
#endif

}  // extern "C"
