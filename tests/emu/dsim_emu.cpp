// dsim_emu.cpp -- TEST-ONLY harness: runs the kernel phase code of diffrl_amd/csrc/dsim_core.hpp with a
// lane-serial executor on the host so that the phase logic (indexing, adjoint algebra) can be
// unit-tested in the GPU-less build container.  It is NOT a CPU fallback: nothing in diffrl_amd/
// loads it, and it is built only by tests/emu/Makefile.  The GPU tests (-m gpu) exercise the real
// HIP kernels through the C ABI.
#include <ucontext.h>

#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define DSIM_FN static inline
#include "../../diffrl_amd/csrc/dsim_core.hpp"
#include "../../diffrl_amd/csrc/dsim_literal.hpp"
#ifdef DSIM_STATIC_LAYOUTS_FILE   // (a generated header with user models: tests/inject/dsim_static_layouts_user.hpp, see the Makefile)
#include DSIM_STATIC_LAYOUTS_FILE
#else
#include "../../diffrl_amd/csrc/dsim_static_layouts.hpp"
#endif

// NW wavefronts per environment: NL = 64 * NW lanes (the library's kernels use 1 or 4).
//
// One wavefront (NW == 1): the lanes of a phase run as 64 coroutines in lock step, so that the phase code may use the
// wavefront's cross-lane primitives -- shfl (ds_bpermute on the GPU), bcast (v_readlane), lds_fence -- exactly where the
// kernels use them: a lane that reaches a collective deposits its value and yields; once every lane still running has
// arrived, each continues with the value of its source lane.  Four wavefronts: plain lane-serial execution, no
// cross-lane primitives (the kernels have none across wavefronts either; WAVE_OPS is false).
// LANES == 32: the lane count the phase code sees when TWO environments share a wavefront (dsim_hip.hip: DSIM_MODE_PAIR,
// DevExec EPW == 2) -- every variant choice that depends on Exec::NL is then the pair kernels'.
template <int NW, int LANES = DSIM_NL> struct HostExecT {
    static constexpr int NL = LANES * NW;
    static constexpr bool WAVE_OPS = NW == 1;
    static constexpr size_t STACK = 512 * 1024;
    ucontext_t main_ctx_, lane_ctx_[NL];
    std::vector<char> stacks_;
    std::function<void(int)> body_;
    bool done_[NL];
    int cur_ = 0, parity_[NL];
    float slot_[2][NL];
    static HostExecT*& self() {
        static thread_local HostExecT* p = nullptr;
        return p;
    }
    static void tramp(int lane) {
        HostExecT* e = self();
        e->body_(lane);
        e->done_[lane] = true;
        swapcontext(&e->lane_ctx_[lane], &e->main_ctx_);
    }
    // the first `n` lanes as lock-step coroutines (cross-lane primitives allowed inside f)
    template <class F> void run_coro(F&& f, int n) {
        if (stacks_.size() < STACK * (size_t)n) stacks_.resize(STACK * (size_t)n);
        body_ = [&f](int lane) { f(lane); };
        self() = this;
        for (int lane = 0; lane < n; ++lane) {
            done_[lane] = false;
            parity_[lane] = 0;
            getcontext(&lane_ctx_[lane]);
            lane_ctx_[lane].uc_stack.ss_sp = stacks_.data() + STACK * lane;
            lane_ctx_[lane].uc_stack.ss_size = STACK;
            lane_ctx_[lane].uc_link = &main_ctx_;
            makecontext(&lane_ctx_[lane], (void (*)())tramp, 1, lane);
        }
        for (int left = n; left > 0;) {
            left = 0;
            for (int lane = 0; lane < n; ++lane) {
                if (done_[lane]) continue;
                cur_ = lane;
                swapcontext(&main_ctx_, &lane_ctx_[lane]);  // runs until the lane's next collective or its end
                if (!done_[lane]) ++left;
            }
        }
    }
    template <class F> void run(F&& f) {
        if constexpr (!WAVE_OPS) {
            for (int lane = 0; lane < NL; ++lane) f(lane);
        } else {
            run_coro(f, NL);
        }
    }
    // a phase of the environment's FIRST wavefront alone (device: DevExec::run_wave0), with its cross-lane primitives
    static constexpr bool WAVE0_OPS = true;
    template <class F> void run_wave0(F&& f) { run_coro(f, LANES); }
    // collectives (only meaningful inside run(), NW == 1).  Every lane that is still running must call the same sequence.
    void arrive() {
        const int lane = cur_;
        swapcontext(&lane_ctx_[lane], &main_ctx_);  // back to the scheduler; resumed when all running lanes have arrived
    }
    float shfl(float v, int src) {
        const int lane = cur_, p = parity_[lane];
        slot_[p][lane] = v;
        parity_[lane] ^= 1;
        arrive();
        return slot_[p][src & (NL - 1)];
    }
    float bcast(float v, int src) { return shfl(v, src); }
    // DPP row shift of the device executor: lane + D of the same 16-lane row, 0 beyond the row's end
    template <int D> float from_above(float v) {
        const int lane = cur_, src = lane + D;
        const float r = shfl(v, src < LANES ? src : lane);
        return (src < LANES && (src >> 4) == (lane >> 4)) ? r : 0.f;
    }
    template <int D> float from_below(float v) {
        const int lane = cur_, src = lane - D;
        const float r = shfl(v, src >= 0 ? src : lane);
        return (src >= 0 && (src >> 4) == (lane >> 4)) ? r : 0.f;
    }
    template <bool FIRST = true> void add_from_next(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        const int lane = cur_;
        auto nx = [&](float v) { const float r = shfl(v, lane + 1 < LANES ? lane + 1 : lane); return lane + 1 < LANES ? r : 0.f; };
        const float y0 = nx(a0), y1 = nx(a1), y2 = nx(a2), y3 = nx(a3), y4 = nx(a4), y5 = nx(a5);
        a0 = __builtin_fmaf(y0, w, a0); a1 = __builtin_fmaf(y1, w, a1); a2 = __builtin_fmaf(y2, w, a2);
        a3 = __builtin_fmaf(y3, w, a3); a4 = __builtin_fmaf(y4, w, a4); a5 = __builtin_fmaf(y5, w, a5);
    }
    template <int SRC> void add_from_lane(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        const float s0 = bcast(a0, SRC), s1 = bcast(a1, SRC), s2 = bcast(a2, SRC), s3 = bcast(a3, SRC), s4 = bcast(a4, SRC), s5 = bcast(a5, SRC);
        a0 = __builtin_fmaf(s0, w, a0); a1 = __builtin_fmaf(s1, w, a1); a2 = __builtin_fmaf(s2, w, a2);
        a3 = __builtin_fmaf(s3, w, a3); a4 = __builtin_fmaf(s4, w, a4); a5 = __builtin_fmaf(s5, w, a5);
    }
    void loads_landed() {}
    void group_sync() {}                                       // (no helper wavefront here: DsimHelperCommit is off)
    void helper_prefetch(const float*, int) {}
    void helper_commit(float*, int, const float*) {}
    void helper_prefetch_aux(const float*, int) {}
    void helper_commit_aux(float*, float*, int) {}
    template <int D, bool FIRST = true> void add_from_above(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float w) {
        const float y0 = from_above<D>(a0), y1 = from_above<D>(a1), y2 = from_above<D>(a2), y3 = from_above<D>(a3),
                    y4 = from_above<D>(a4), y5 = from_above<D>(a5);
        a0 = __builtin_fmaf(y0, w, a0); a1 = __builtin_fmaf(y1, w, a1); a2 = __builtin_fmaf(y2, w, a2);
        a3 = __builtin_fmaf(y3, w, a3); a4 = __builtin_fmaf(y4, w, a4); a5 = __builtin_fmaf(y5, w, a5);
    }
    void lds_fence() { arrive(); }
    void system_fence() {}
    template <class F> void fire(F&& f) { run(f); }
    // the helper wavefront of the device executor (dsim_hip.hip) does not exist here: both blocks of a split phase run in
    // the one emulated wave, one after the other
    static constexpr bool HAS_HELPER = false;
    template <class F> void run_both(F&& f) { run(f); }
    template <class FM, class FH> void fork_join(FM&& fm, FH&& fh) {
        run([&](int lane) {
            fm(lane);
            fh(lane);
        });
    }
    // fm hands over to fh at its mid() call (device: the helper's barrier): every lane passes mid() before any lane gets to fh
    template <class FM, class FH> void fork_join_mid(FM&& fm, FH&& fh) { fork_join(fm, fh); }
    template <class FM, class FH> void fork_mid_detached(FM&& fm, FH&& fh) { fork_join(fm, fh); }
    // Several wavefronts (device: DevExec::fork_wave0): the first wavefront's lanes and the others' run as coroutines of one group;
    // mid() / mid2() / side_done_w() are WORKGROUP BARRIERS there -- no lane passes one before every lane of the group has reached
    // it -- while the first wavefront's cross-lane primitives (shfl, lds_fence) stay lock-step points of its own lanes only (the
    // other lanes' blocks use none).
    int bar_n_ = 0, bar_cnt_ = 0, bar_gen_ = 0;
    void wbar() {
        const int gen = bar_gen_;
        if (++bar_cnt_ == bar_n_) {
            bar_cnt_ = 0;
            ++bar_gen_;
        }
        do arrive();
        while (bar_gen_ == gen);
    }
    template <class F0, class FR> void fork_wave0(F0&& f0, FR&& fr) {
        static_assert(NW > 1, "fork_wave0 belongs to the mapping with several wavefronts per environment");
        bar_n_ = NL;
        bar_cnt_ = 0;
        in_coro_ = true;
        run_coro([&](int lane) {
            if (lane < LANES) f0(lane);
            else fr(lane - LANES);
        }, NL);
        in_coro_ = false;
        bar_n_ = 0;
    }
    void mid() {
        if (bar_n_) wbar();
        else arrive();
    }
    void mid2() {
        if (bar_n_) wbar();
    }
    void side_done_w() { wbar(); }
    // side block first, then the main block, whose side_done() is a point every lane passes (device: the helper's barrier)
    template <class FM, class FH> void fork_side(FM&& fm, FH&& fh) {
        run([&](int lane) {
            fh(lane);
            arrive();
            fm(lane);
        });
    }
    void side_done() { arrive(); }
    void stamp() {}
    // host form of the register / v_readlane Gauss-Jordan of the kernels (dsim_hip.hip: dsim_wave_gj, deferred scaling of the
    // pivot rows): same formulas in the same order (the device's pivot reciprocal is v_rcp_f32 + a Newton step, here 1 / x)
    template <int N> void wave_gj(float* H) {
        float scale[N];
        for (int i = 0; i < N; ++i) scale[i] = 1.0f;
        for (int k = 0; k < N; ++k) {
            const float rp = 1.0f / H[k * N + k];
            float f[N];
            for (int i = 0; i < N; ++i) f[i] = (i == k) ? 0.0f : H[i * N + k] * rp;
            scale[k] = rp;
            for (int i = 0; i < N; ++i) H[i * N + k] = (i == k) ? 1.0f : 0.0f;
            float hk[N];
            for (int j = 0; j < N; ++j) hk[j] = H[k * N + j];
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) H[i * N + j] = __builtin_fmaf(-f[i], hk[j], H[i * N + j]);
        }
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) H[i * N + j] *= scale[i];
    }
    static constexpr bool WAVE_GJ_PAD = true;
    template <int NB> void wave_gj_pad(float* H, int n) {   // (device: the matrix padded with an identity block in registers)
        float P[NB * NB];
        for (int i = 0; i < NB; ++i)
            for (int j = 0; j < NB; ++j) P[i * NB + j] = (i < n && j < n) ? H[i * n + j] : (i == j ? 1.0f : 0.0f);
        wave_gj<NB>(P);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) H[i * n + j] = P[i * NB + j];
    }
    void mark(int) {}
    void begin_request() {}
    void begin() {}   // the host image is set up by make_ctx (constants copied, work area zero)
    float io_[NL][DSIM_IO_MAX];   // early-load registers of the specialised kernels
    float* io(int lane) { return io_[lane]; }
    float hacc_[NL][DSIM_HACC_MAX];  // what a lane keeps in registers across phases on the GPU
    float* hacc(int lane) { return hacc_[lane]; }
    float hpf_[NL][DSIM_HPF_MAX];
    float* hpf(int lane) { return hpf_[lane]; }
    DsimTopoRegs topo_[NL];
    DsimTopoRegs& topo(int lane) { return topo_[lane]; }
    const float* pf_src = nullptr;
    void prefetch(const float* row, int) { pf_src = row; }
    void commit(float* dst, int words, int lane) {
        for (int k = lane; k < words; k += NL) dst[k] = pf_src[k];
    }
    // (per lane: on the device the row waits in each lane's own registers; here a lane of fork_wave0's second block requests the
    // row after the next while the lanes behind it have not committed theirs yet)
    const float* pf_rest_[NL] = {};
    bool in_coro_ = false;
    void prefetch_rest(const float* row, int) {
        if (in_coro_) pf_rest_[cur_] = row;
        else for (int l = 0; l < NL; ++l) pf_rest_[l] = row;
    }
    void commit_rest(float* dst, int words, int lane) {
        for (int k = lane; k < words; k += NL - LANES) dst[k] = pf_rest_[lane + LANES][k];
    }
};
typedef HostExecT<1> HostExec;
static int g_lean = 0;   // checkpoint mode of the emu entry points (dsim_model_set_ckpt_mode of the library)
extern "C" void dsim_emu_set_ckpt_lean(int on) { g_lean = on; }
static int emu_row(const DsimLayout& lay) { return g_lean ? lay.o.xsc - lay.o.q : lay.o.save_words; }
static int g_waves = 1;
extern "C" void dsim_emu_set_waves(int w) { g_waves = w > 1 ? 4 : 1; }
static int g_half = 0;   // 32 lanes per environment (specialised one-wave variants only: the pair kernels)
extern "C" void dsim_emu_set_half_wave(int on) { g_half = on; }

static void make_ctx(const DsimLayout& lay, std::vector<float>& lds, DsimCtx& c, float h) {
    lds.assign(lay.o.total_words, 0.f);
    memcpy(lds.data(), lay.cblob.data(), sizeof(uint32_t) * lay.o.const_words);
    c.s = lds.data(); c.k = c.s;
    c.o = lay.o;
    c.d = lay.d;
    c.h = h;
}

extern "C" {

int dsim_emu_layout(const dsim_model_desc* m, int* off_out, int n_off, int* dims_out) {
    DsimLayout lay;
    std::string err = dsim_build_layout(*m, lay);
    if (!err.empty()) return -1;
    const int n = (int)(sizeof(DsimOff) / sizeof(int));
    if (n_off < n) return n;
    memcpy(off_out, &lay.o, sizeof(DsimOff));
    memcpy(dims_out, &lay.d, sizeof(DsimDims));
    return n;
}

// one substep (with mass-matrix refresh) for ONE env; returns the LDS image for inspection
int dsim_emu_substep_image(const dsim_model_desc* m, const float* q, const float* qd, const float* act,
                           const float* mact, float h, float* image) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    std::vector<float> lds;
    DsimCtx c;
    make_ctx(lay, lds, c, h);
    memcpy(lds.data() + lay.o.q, q, 4 * lay.d.nq);
    memcpy(lds.data() + lay.o.qd, qd, 4 * lay.d.nd);
    memcpy(lds.data() + lay.o.act, act, 4 * lay.d.nd);
    if (lay.d.M) memcpy(lds.data() + lay.o.mact, mact, 4 * lay.d.M);
    HostExec ex;
    dsim_fwd_kinematics(c, ex);
    dsim_fwd_external(c, ex);
    dsim_fwd_tau(c, ex);
    dsim_fwd_mass(c, ex);
    dsim_fwd_solve(c, ex);
    memcpy(image, lds.data(), 4 * lay.o.total_words);
    return 0;
}

int dsim_emu_step_forward(const dsim_model_desc* m, int n_envs, const float* q_in, const float* qd_in,
                          const float* act, const float* mact, float dt, int substeps, int mm_freq, float* q_out,
                          float* qd_out, float* ckpt) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    const int nq = lay.d.nq, nd = lay.d.nd, M = lay.d.M;
    HostExec ex;
    for (int e = 0; e < n_envs; ++e) {
        std::vector<float> lds;
        DsimCtx c;
        make_ctx(lay, lds, c, dt / float(substeps));
        dsim_sim_step_forward(c, ex, substeps, mm_freq, q_in + (size_t)e * nq, qd_in + (size_t)e * nd,
                              act + (size_t)e * nd, M ? mact + (size_t)e * M : nullptr, q_out + (size_t)e * nq,
                              qd_out + (size_t)e * nd, ckpt ? ckpt + (size_t)e * dsim_ckpt_words(lay.o.save_words, nq, nd, substeps, mm_freq) : nullptr);
    }
    return 0;
}
}

extern "C" int dsim_emu_step_backward(const dsim_model_desc* m, int n_envs, const float* ckpt, const float* act,
                                      const float* mact, float dt, int substeps, int mm_freq, const float* gq_out,
                                      const float* gqd_out, float* gq_in, float* gqd_in, float* gact, float* gmact) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    const int nq = lay.d.nq, nd = lay.d.nd, M = lay.d.M;
    HostExec ex;
    for (int e = 0; e < n_envs; ++e) {
        std::vector<float> lds;
        DsimCtx c;
        make_ctx(lay, lds, c, dt / float(substeps));
        dsim_sim_step_backward(c, ex, substeps, mm_freq, ckpt + (size_t)e * dsim_ckpt_words(lay.o.save_words, nq, nd, substeps, mm_freq), act + (size_t)e * nd,
                               M ? mact + (size_t)e * M : nullptr, gq_out + (size_t)e * nq, gqd_out + (size_t)e * nd,
                               gq_in + (size_t)e * nq, gqd_in + (size_t)e * nd, gact ? gact + (size_t)e * nd : nullptr,
                               (gmact && M) ? gmact + (size_t)e * M : nullptr);
    }
    return 0;
}

// dsim_step_backward_literal on the host: the adjoint phases with the export of (gq_1, gqd_1, adj H), then the forward-mode
// tangent of the first substep (diffrl_amd/csrc/dsim_literal.hpp) link by link, exactly what dsim_literal_radial_kernel does
extern "C" int dsim_emu_step_backward_literal(const dsim_model_desc* m, int n_envs, const float* ckpt, const float* act,
                                              const float* mact, float dt, int substeps, int mm_freq, const float* gq_out,
                                              const float* gqd_out, float* gq_in, float* gqd_in, float* gact, float* gmact,
                                              float* lit_out) {   // lit_out: [n_envs][nq + nd + nd * nd] or null (tests)
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    const int nq = lay.d.nq, nd = lay.d.nd, M = lay.d.M, L = lay.d.L;
    if (L > DSIM_LIT_LMAX || nd > DSIM_LIT_NDMAX) return -4;
    HostExec ex;
    const size_t stride = dsim_ckpt_words(lay.o.save_words, nq, nd, substeps, mm_freq);
    std::vector<float> lit((size_t)nq + nd + (size_t)nd * nd);
    dsim_lit::Consts cc;
    cc.cb = lay.cblob.data();
    cc.o = lay.o;
    cc.d = lay.d;
    for (int e = 0; e < n_envs; ++e) {
        std::vector<float> lds;
        DsimCtx c;
        make_ctx(lay, lds, c, dt / float(substeps));
        const float* row = ckpt + (size_t)e * stride;
        dsim_sim_step_backward(c, ex, substeps, mm_freq, row, act + (size_t)e * nd, M ? mact + (size_t)e * M : nullptr,
                               gq_out + (size_t)e * nq, gqd_out + (size_t)e * nd, gq_in + (size_t)e * nq, gqd_in + (size_t)e * nd,
                               gact ? gact + (size_t)e * nd : nullptr, (gmact && M) ? gmact + (size_t)e * M : nullptr, lit.data());
        const float* hinv = row + (size_t)substeps * lay.o.save_words;
        if (lit_out) memcpy(lit_out + (size_t)e * lit.size(), lit.data(), sizeof(float) * lit.size());
        for (int i = 0; i < L; ++i) {
            const int type = cc.I(lay.o.jtype, i);
            if (type != DSIM_JOINT_BALL && type != DSIM_JOINT_FREE) continue;
            const float rho = dsim_lit::radial(cc, row, row + (lay.o.qd - lay.o.q), act + (size_t)e * nd, M ? mact + (size_t)e * M : nullptr,
                                               dt / float(substeps), hinv, lit.data(), lit.data() + nq, lit.data() + nq + nd, i);
            const int qs = cc.I(lay.o.qstart, i) + (type == DSIM_JOINT_FREE ? 3 : 0);
            for (int j = 0; j < 4; ++j) gq_in[(size_t)e * nq + qs + j] += rho * row[qs + j];
        }
    }
    return 0;
}

// test helper: tangents (d tau [nd], d qdd [nd], d H [nd][nd]) of the first substep along the radial direction of link `jl`'s quaternion
extern "C" int dsim_emu_literal_tangents(const dsim_model_desc* m, const float* q, const float* qd, const float* act, const float* mact,
                                         float h, const float* hinv, int jl, float* out, const float* lit, float* rho_out) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    const int nq = lay.d.nq, nd = lay.d.nd;
    dsim_lit::Consts cc;
    cc.cb = lay.cblob.data();
    cc.o = lay.o;
    cc.d = lay.d;
    std::vector<float> z((size_t)nq + nd + (size_t)nd * nd, 0.f);
    for (int k = 0; k < 2 * nd + nd * nd; ++k) out[k] = 0.f;
    const float* l = lit ? lit : z.data();
    const float rho = dsim_lit::radial(cc, q, qd, act, mact, h, hinv, l, l + nq, l + nq + nd, jl, out);
    if (rho_out) *rho_out = rho;
    return 0;
}

// ---- fused env surface (host lane-serial) ----
static DsimEnvSpec to_spec(const dsim_env_spec* e) {
    DsimEnvSpec sp;
    sp.kind = e->kind; sp.rew_kind = e->rew_kind; sp.n_act = e->n_act; sp.n_obs = e->n_obs;
    sp.act_offset = e->act_offset; sp.act_muscle = e->act_muscle; sp.obs_actions = e->obs_actions; sp.sanitize = e->sanitize_grads;
    for (int k = 0; k < 4; ++k) { sp.isr[k] = e->inv_start_rot[k]; sp.pen[k] = e->cartpole_penalties[k]; }
    sp.tgt_x = e->target_x; sp.tgt_z = e->target_z; sp.term_h = e->termination_height;
    sp.term_tol = e->termination_tolerance; sp.h_scale = e->height_rew_scale; sp.act_pen = e->action_penalty;
    sp.vel_scale = e->joint_vel_obs_scaling; sp.act_scale = e->act_scale;
    return sp;
}

// The per-model specialised code paths (compile-time layouts: `if constexpr (DsimIsStatic...)` branches, bounded sums)
// can be exercised on the host too: with dsim_emu_use_static(1) the env entry points below instantiate the phase code
// with the generated all-constexpr layout of the matching model, exactly as the library's dispatch() does.
static int g_use_static = 0;
extern "C" void dsim_emu_use_static(int on) { g_use_static = on; }

template <class F, class O, class D> static int emu_waves(F&& f, O o, D d) {
    if (g_waves > 1) {
        static HostExecT<4> ex4;
        return g_lean ? f(o, d, ex4, std::true_type{}) : f(o, d, ex4, std::false_type{});
    }
    if constexpr (dsim_pair_ok<D>()) {
        if (g_half) {
            static HostExecT<1, 32> exh;
            return g_lean ? f(o, d, exh, std::true_type{}) : f(o, d, exh, std::false_type{});
        }
    }
    if (g_half) return -3;   // no pair kernels for this model
    static HostExecT<1> ex1;
    return g_lean ? f(o, d, ex1, std::true_type{}) : f(o, d, ex1, std::false_type{});
}
template <class F> static int emu_dispatch(const DsimLayout& lay, F&& f) {
    if (!g_use_static) return emu_waves(f, lay.o, lay.d);
    const size_t no = sizeof(DsimOff) / sizeof(int), nd = sizeof(DsimDims) / sizeof(int);
#define DSIM_EMU_CASE(T)                                                                                            \
    if (sizeof(kDsimStatic##T) == (no + nd) * sizeof(int) && memcmp(kDsimStatic##T, &lay.o, no * sizeof(int)) == 0 && \
        memcmp(kDsimStatic##T + no, &lay.d, nd * sizeof(int)) == 0)                                                  \
        return emu_waves(f, DsimOff##T{}, DsimDims##T{});
    DSIM_STATIC_VARIANTS(DSIM_EMU_CASE)
#undef DSIM_EMU_CASE
    return -2;  // no specialised variant for this model
}

static DsimEpisode to_episode(const dsim_episode* episode) {
    DsimEpisode ep{};
    if (episode) {
        ep.progress = reinterpret_cast<long long*>(episode->progress);
        ep.done = reinterpret_cast<long long*>(episode->done);
        ep.obs_before = episode->obs_before_reset;
        ep.reset_q = episode->reset_q; ep.reset_qd = episode->reset_qd; ep.reset_count = episode->reset_count;
        ep.pool = episode->reset_pool; ep.episode_length = episode->episode_length;
        ep.height_terminate = episode->height_terminate; ep.check_invalid = episode->check_invalid;
        ep.noise_q = episode->noise_q; ep.noise_qd = episode->noise_qd; ep.noise_angle = episode->noise_angle;
        ep.seed = episode->seed;
    }
    return ep;
}

extern "C" int dsim_emu_env_forward(const dsim_model_desc* m, const dsim_env_spec* env, int n_envs, const float* q_in,
                                    const float* qd_in, const float* actions, float dt, int substeps, int mm_freq,
                                    float* q_out, float* qd_out, float* obs, float* rew, float* ckpt,
                                    const dsim_episode* episode) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    const int nq = lay.d.nq, nd = lay.d.nd;
    DsimEnvSpec sp = to_spec(env);
    DsimEpisode ep = to_episode(episode);
    const size_t stride = dsim_ckpt_words(emu_row(lay), nq, nd, substeps, mm_freq);
    return emu_dispatch(lay, [&](auto o, auto d, auto& ex, auto lean) {
        for (int e = 0; e < n_envs; ++e) {
            std::vector<float> lds(lay.o.total_words, 0.f);
            memcpy(lds.data(), lay.cblob.data(), sizeof(uint32_t) * lay.o.const_words);
            DsimCtxT<decltype(o), decltype(d), decltype(lean)::value> c;
            c.s = lds.data(); c.k = c.s; c.o = o; c.d = d; c.h = dt / float(substeps);
            dsim_env_fused_forward(c, ex, sp, substeps, mm_freq, q_in + (size_t)e * nq, qd_in + (size_t)e * nd,
                                   actions + (size_t)e * sp.n_act, q_out + (size_t)e * nq, qd_out + (size_t)e * nd,
                                   obs + (size_t)e * sp.n_obs, rew + e, ckpt ? ckpt + (size_t)e * stride : nullptr, ep, e,
                                   n_envs);
        }
        return 0;
    });
}

extern "C" int dsim_emu_env_backward(const dsim_model_desc* m, const dsim_env_spec* env, int n_envs, const float* ckpt,
                                     const float* actions, float dt, int substeps, int mm_freq, const float* gq_out,
                                     const float* gqd_out, const float* gobs, const float* grew,
                                     const float* gobs_before, float* gq_in, float* gqd_in, float* gactions) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    const int nq = lay.d.nq, nd = lay.d.nd;
    DsimEnvSpec sp = to_spec(env);
    const size_t stride = dsim_ckpt_words(emu_row(lay), nq, nd, substeps, mm_freq);
    return emu_dispatch(lay, [&](auto o, auto d, auto& ex, auto lean) {
        for (int e = 0; e < n_envs; ++e) {
            std::vector<float> lds(lay.o.total_words, 0.f);
            memcpy(lds.data(), lay.cblob.data(), sizeof(uint32_t) * lay.o.const_words);
            DsimCtxT<decltype(o), decltype(d), decltype(lean)::value> c;
            c.s = lds.data(); c.k = c.s; c.o = o; c.d = d; c.h = dt / float(substeps);
            dsim_env_fused_backward(c, ex, sp, substeps, mm_freq, ckpt + (size_t)e * stride,
                                    actions + (size_t)e * sp.n_act, gq_out ? gq_out + (size_t)e * nq : nullptr,
                                    gqd_out ? gqd_out + (size_t)e * nd : nullptr,
                                    gobs ? gobs + (size_t)e * sp.n_obs : nullptr, grew ? grew + e : nullptr,
                                    gobs_before ? gobs_before + (size_t)e * sp.n_obs : nullptr, gq_in + (size_t)e * nq,
                                    gqd_in + (size_t)e * nd, gactions + (size_t)e * sp.n_act);
        }
        return 0;
    });
}

extern "C" long long dsim_emu_ckpt_floats(const dsim_model_desc* m, int substeps, int mm_freq) {
    DsimLayout lay;
    if (!dsim_build_layout(*m, lay).empty()) return -1;
    return dsim_ckpt_words(emu_row(lay), lay.d.nq, lay.d.nd, substeps, mm_freq);
}
