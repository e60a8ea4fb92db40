// dsim_oracle.cpp -- TEST INFRASTRUCTURE ONLY (the checker, never the product).
//
// Scalar CPU restatement of the articulated rigid-body substep of NVlabs/DiffRL's dflex and of its
// reverse-mode adjoint.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
// load this library; diffrl_amd/ must never import it (the product path is the HIP library and
// fails loudly when that is missing).
//
// Parity status: PINNED by tests/golden/*.npz, which oracle/gen_golden.py generated from the real
// reference running in the build container (the reference itself has no golden vectors / KATs for
// this path).  tests/test_oracle_golden.py holds this file to those vectors.
//
// Forward: one env at a time, fp32, in the reference's exact operation order:
//   FK          dflex/dflex/sim.py:1638-1711  (compute_link_transform, jcalc_transform 1269-1319)
//   ID          sim.py:1716-1789, 1845-1893   (jcalc_motion 1323-1387, twist/wrench/inertia 1076-1134)
//   contacts    sim.py:1137-1206              (eval_rigid_contacts_art)
//   muscles     sim.py:1209-1265
//   tau         sim.py:1792-1842, 1896-1948   (jcalc_tau 1421-1502)
//   J, M        dflex/dflex/spatial.h:691-738, 801-815
//   P=MJ,H=JtP  dflex/dflex/matnn.h:23-99 (dense_gemm), launched sim.py:2514-2545
//   chol, solve matnn.h:140-230
//   integrate   sim.py:1505-1636, 2052-2081
//   primitives  vec3.h, quat.h:44-121, mat33.h, spatial.h:6-161,166-357,425-586
// Adjoint: instead of transcribing the reference's generated adjoint code, the forward is
// templated on the scalar type and re-run on a tape-recording scalar (operator-overloading
// reverse-mode AD).  That is what the reference's source-to-source transformer does mechanically
// (dflex/dflex/adjoint.py:680-1157), with these reference-specific rules encoded explicitly:
//   min/max: gradient to the selected argument, ties to the 2nd        (adjoint.h:129-143)
//   step():  zero gradient                                              (adjoint.h:177-180)
//   length/normalize of a zero vector: zero gradient                    (vec3.h:194-222, quat.h:182-192)
//   Cholesky adjoint is a no-op; d qdd/d H goes through dense_solve:
//     t = (LL^T)^-1 adj_qdd ; adj_tau += t ; adj_H[i][j] += -t_i*qdd_j  (matnn.h:288-336)
//   a stale H reused on non-update substeps keeps accumulating adj_H    (adjoint.py:2171-2186)
//   non-penetrating contacts return before contributing anything        (sim.py:1179-1180)
// The hand-derived HIP adjoint is therefore checked against an independently derived one.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/dsim.h"

namespace {

// ------------------------------------------------------------------------------------------------
// tape
struct Node {
    int p[4];
    float d[4];
};
struct SolveBlock {
    int n;
    int end;                 // tape position right after the outputs were created
    std::vector<float> L;    // Cholesky factor values (constant)
    std::vector<int> H;      // var ids of H (n*n)
    std::vector<int> tau;    // var ids of rhs (n)
    std::vector<int> x;      // var ids of solution (n)
    std::vector<float> xv;   // forward solution values
};
struct Tape {
    std::vector<Node> nodes;
    std::vector<SolveBlock> blocks;
    void clear() { nodes.clear(); blocks.clear(); }
};
thread_local Tape* g_tape = nullptr;

inline int push(int p0, float d0, int p1 = -1, float d1 = 0.f, int p2 = -1, float d2 = 0.f, int p3 = -1, float d3 = 0.f) {
    Node n;
    n.p[0] = p0; n.p[1] = p1; n.p[2] = p2; n.p[3] = p3;
    n.d[0] = d0; n.d[1] = d1; n.d[2] = d2; n.d[3] = d3;
    g_tape->nodes.push_back(n);
    return (int)g_tape->nodes.size() - 1;
}

struct Var {
    float v;
    int i;  // -1: constant
    Var() : v(0.f), i(-1) {}
    Var(float c) : v(c), i(-1) {}
    Var(float c, int id) : v(c), i(id) {}
};
inline Var leaf(float v) { return Var(v, push(-1, 0.f)); }

inline Var operator+(Var a, Var b) {
    if (a.i < 0 && b.i < 0) return Var(a.v + b.v);
    return Var(a.v + b.v, push(a.i, 1.f, b.i, 1.f));
}
inline Var operator-(Var a, Var b) {
    if (a.i < 0 && b.i < 0) return Var(a.v - b.v);
    return Var(a.v - b.v, push(a.i, 1.f, b.i, -1.f));
}
inline Var operator*(Var a, Var b) {
    if (a.i < 0 && b.i < 0) return Var(a.v * b.v);
    return Var(a.v * b.v, push(a.i, b.v, b.i, a.v));
}
inline Var operator/(Var a, Var b) {
    if (a.i < 0 && b.i < 0) return Var(a.v / b.v);
    // adjoint.h:185 adj_div: adj_a += r/b ; adj_b -= r*(a/b)/b
    return Var(a.v / b.v, push(a.i, 1.f / b.v, b.i, -(a.v / b.v) / b.v));
}
inline Var operator-(Var a) {
    if (a.i < 0) return Var(-a.v);
    return Var(-a.v, push(a.i, -1.f));
}
inline Var& operator+=(Var& a, Var b) { a = a + b; return a; }
inline Var& operator-=(Var& a, Var b) { a = a - b; return a; }
inline bool operator<(Var a, Var b) { return a.v < b.v; }
inline bool operator>(Var a, Var b) { return a.v > b.v; }
inline bool operator>=(Var a, Var b) { return a.v >= b.v; }

inline float val(float x) { return x; }
inline float val(Var x) { return x.v; }

inline float s_sqrt(float x) { return sqrtf(x); }
inline Var s_sqrt(Var x) {
    float r = sqrtf(x.v);
    if (x.i < 0) return Var(r);
    return Var(r, push(x.i, 0.5f * (1.0f / r)));
}
inline float s_sin(float x) { return sinf(x); }
inline float s_cos(float x) { return cosf(x); }
inline Var s_sin(Var x) { return x.i < 0 ? Var(sinf(x.v)) : Var(sinf(x.v), push(x.i, cosf(x.v))); }
inline Var s_cos(Var x) { return x.i < 0 ? Var(cosf(x.v)) : Var(cosf(x.v), push(x.i, -sinf(x.v))); }
// adjoint.h:94-95, 129-143
inline float s_min(float a, float b) { return a < b ? a : b; }
inline Var s_min(Var a, Var b) { return a.v < b.v ? a : b; }
// adjoint.h:99 step(x) = x < 0 ? 1 : 0, no gradient
template <class T> inline float s_step(T x) { return val(x) < 0.0f ? 1.0f : 0.0f; }

// ------------------------------------------------------------------------------------------------
// vec3 / quat / transform / spatial types, templated on the scalar
template <class T> struct V3 {
    T x, y, z;
    V3() : x(0.f), y(0.f), z(0.f) {}
    V3(T a, T b, T c) : x(a), y(b), z(c) {}
};
template <class T> inline V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> inline V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> inline V3<T> operator*(V3<T> a, T s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> inline V3<T> operator/(V3<T> a, T s) { return {a.x / s, a.y / s, a.z / s}; }
template <class T> inline T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> inline V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// vec3.h:65-68 / 194-197: length, zero gradient at |a| == 0
inline float length(V3<float> a) { return sqrtf(dot(a, a)); }
inline Var length(V3<Var> a) {
    float l = sqrtf(a.x.v * a.x.v + a.y.v * a.y.v + a.z.v * a.z.v);
    if (l > 0.0f) return Var(l, push(a.x.i, a.x.v / l, a.y.i, a.y.v / l, a.z.i, a.z.v / l));
    return Var(l);
}
// vec3.h:70-77
template <class T> inline V3<T> normalize(V3<T> a) {
    T l = length(a);
    if (val(l) > 0.0f) return a / l;
    return V3<T>();
}

template <class T> struct Q4 {
    T x, y, z, w;
    Q4() : x(0.f), y(0.f), z(0.f), w(0.f) {}
    Q4(T a, T b, T c, T d) : x(a), y(b), z(c), w(d) {}
};
template <class T> inline Q4<T> quat_identity() { return Q4<T>(T(0.f), T(0.f), T(0.f), T(1.f)); }
// quat.h:44-52
template <class T> inline Q4<T> quat_from_axis_angle(V3<T> axis, T angle) {
    T half = angle * T(0.5f);
    T w = s_cos(half);
    T s = s_sin(half);
    V3<T> v = axis * s;
    return Q4<T>(v.x, v.y, v.z, w);
}
// quat.h:100-106
template <class T> inline Q4<T> qmul(Q4<T> a, Q4<T> b) {
    return Q4<T>(a.w * b.x + b.w * a.x + a.y * b.z - b.y * a.z,
                 a.w * b.y + b.w * a.y + a.z * b.x - b.z * a.x,
                 a.w * b.z + b.w * a.z + a.x * b.y - b.x * a.y,
                 a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
// quat.h:113-116
template <class T> inline V3<T> rotate(Q4<T> q, V3<T> x) {
    V3<T> qv(q.x, q.y, q.z);
    return x * (T(2.0f) * q.w * q.w - T(1.0f)) + cross(qv, x) * q.w * T(2.0f) + qv * dot(qv, x) * T(2.0f);
}
// quat.h:70-83
template <class T> inline Q4<T> qnormalize(Q4<T> q) {
    T l = s_sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    if (val(l) > 0.0f) {
        T inv_l = T(1.0f) / l;
        return Q4<T>(q.x * inv_l, q.y * inv_l, q.z * inv_l, q.w * inv_l);
    }
    return quat_identity<T>();
}

template <class T> struct Xf {
    V3<T> p;
    Q4<T> q;
};
template <class T> inline Xf<T> xf_identity() { return Xf<T>{V3<T>(), quat_identity<T>()}; }
// spatial.h:190-193
template <class T> inline Xf<T> xf_mul(Xf<T> a, Xf<T> b) { return Xf<T>{rotate(a.q, b.p) + a.p, qmul(a.q, b.q)}; }
// spatial.h:209-212
template <class T> inline V3<T> xf_point(Xf<T> t, V3<T> x) { return t.p + rotate(t.q, x); }

template <class T> struct SV {
    V3<T> w, v;
    SV() {}
    SV(V3<T> a, V3<T> b) : w(a), v(b) {}
    T get(int i) const { return i < 3 ? (i == 0 ? w.x : (i == 1 ? w.y : w.z)) : (i == 3 ? v.x : (i == 4 ? v.y : v.z)); }
    void set(int i, T s) {
        switch (i) {
            case 0: w.x = s; break;
            case 1: w.y = s; break;
            case 2: w.z = s; break;
            case 3: v.x = s; break;
            case 4: v.y = s; break;
            default: v.z = s;
        }
    }
};
template <class T> inline SV<T> operator+(SV<T> a, SV<T> b) { return SV<T>(a.w + b.w, a.v + b.v); }
template <class T> inline SV<T> operator-(SV<T> a, SV<T> b) { return SV<T>(a.w - b.w, a.v - b.v); }
template <class T> inline SV<T> operator*(SV<T> a, T s) { return SV<T>(a.w * s, a.v * s); }
template <class T> inline T spatial_dot(SV<T> a, SV<T> b) { return dot(a.w, b.w) + dot(a.v, b.v); }
// spatial.h:56-70
template <class T> inline SV<T> spatial_cross(SV<T> a, SV<T> b) {
    return SV<T>(cross(a.w, b.w), cross(a.v, b.w) + cross(a.w, b.v));
}
template <class T> inline SV<T> spatial_cross_dual(SV<T> a, SV<T> b) {
    return SV<T>(cross(a.w, b.w) + cross(a.v, b.v), cross(a.w, b.v));
}
// sim.py:1076-1103
template <class T> inline SV<T> transform_twist(Xf<T> t, SV<T> x) {
    V3<T> w = rotate(t.q, x.w);
    V3<T> v = rotate(t.q, x.v) + cross(t.p, w);
    return SV<T>(w, v);
}
template <class T> inline SV<T> transform_wrench(Xf<T> t, SV<T> x) {
    V3<T> v = rotate(t.q, x.v);
    V3<T> w = rotate(t.q, x.w) + cross(t.p, v);
    return SV<T>(w, v);
}

template <class T> struct M6 {
    T d[6][6];
    M6() {
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) d[i][j] = T(0.f);
    }
};
// spatial.h:559-575
template <class T> inline M6<T> m6mul(const M6<T>& a, const M6<T>& b) {
    M6<T> o;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j)
            for (int k = 0; k < 6; ++k) o.d[i][j] += a.d[i][k] * b.d[k][j];
    return o;
}
template <class T> inline M6<T> m6T(const M6<T>& a) {
    M6<T> o;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) o.d[i][j] = a.d[j][i];
    return o;
}
// spatial.h:548-557
template <class T> inline SV<T> m6vec(const M6<T>& a, SV<T> b) {
    SV<T> o;
    for (int i = 0; i < 6; ++i) {
        T s(0.f);
        for (int j = 0; j < 6; ++j) s += a.d[i][j] * b.get(j);
        o.set(i, s);
    }
    return o;
}
// sim.py:1105-1134 (spatial_transform_inverse + spatial_transform_inertia), mat33.h:8-21,136-150,187
template <class T> inline M6<T> transform_inertia(Xf<T> t, const M6<T>& I) {
    Q4<T> q_inv(-t.q.x, -t.q.y, -t.q.z, t.q.w);
    V3<T> p = rotate(q_inv, t.p) * (T(0.0f) - T(1.0f));
    V3<T> r1 = rotate(q_inv, V3<T>(T(1.f), T(0.f), T(0.f)));
    V3<T> r2 = rotate(q_inv, V3<T>(T(0.f), T(1.f), T(0.f)));
    V3<T> r3 = rotate(q_inv, V3<T>(T(0.f), T(0.f), T(1.f)));
    T R[3][3] = {{r1.x, r2.x, r3.x}, {r1.y, r2.y, r3.y}, {r1.z, r2.z, r3.z}};  // column constructor
    T K[3][3] = {{T(0.f), -p.z, p.y}, {p.z, T(0.f), -p.x}, {-p.y, p.x, T(0.f)}};
    T S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T s(0.f);
            for (int k = 0; k < 3; ++k) s += K[i][k] * R[k][j];
            S[i][j] = s;
        }
    M6<T> A;  // spatial_adjoint(R, S), spatial.h:595-620
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A.d[i][j] = R[i][j];
            A.d[i + 3][j + 3] = R[i][j];
            A.d[i + 3][j] = S[i][j];
        }
    return m6mul(m6mul(m6T(A), I), A);
}

// ------------------------------------------------------------------------------------------------
inline Xf<float> load_xf(const float* p) {
    Xf<float> t;
    t.p = V3<float>(p[0], p[1], p[2]);
    t.q = Q4<float>(p[3], p[4], p[5], p[6]);
    return t;
}
template <class T> inline Xf<T> lift_xf(Xf<float> a) {
    Xf<T> t;
    t.p = V3<T>(T(a.p.x), T(a.p.y), T(a.p.z));
    t.q = Q4<T>(T(a.q.x), T(a.q.y), T(a.q.z), T(a.q.w));
    return t;
}
template <class T> inline V3<T> lift_v3(const float* p) { return V3<T>(T(p[0]), T(p[1]), T(p[2])); }

// dense_chol, matnn.h:140-169 (values only: its adjoint is a no-op)
void dense_chol(int n, const float* A, const float* reg, float* L) {
    for (int j = 0; j < n; ++j) {
        float s = A[j * n + j] + reg[j];
        for (int k = 0; k < j; ++k) {
            float r = L[j * n + k];
            s -= r * r;
        }
        s = sqrtf(s);
        const float invS = 1.0f / s;
        L[j * n + j] = s;
        for (int i = j + 1; i < n; ++i) {
            s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s * invS;
        }
    }
}
// dense_subs, matnn.h:188-215
void dense_subs(int n, const float* L, const float* b, float* x) {
    for (int i = 0; i < n; ++i) {
        float s = b[i];
        for (int j = 0; j < i; ++j) s -= L[i * n + j] * x[j];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        float s = x[i];
        for (int j = i + 1; j < n; ++j) s -= L[j * n + i] * x[j];
        x[i] = s / L[i * n + i];
    }
}

// plain float solve
inline void dense_solve(int n, const std::vector<float>& L, const std::vector<float>& /*H*/,
                        const std::vector<float>& tau, std::vector<float>& x) {
    x.assign(n, 0.f);
    dense_subs(n, L.data(), tau.data(), x.data());
}
// taped solve: custom block with the reference's adjoint rule (matnn.h:310-336)
inline void dense_solve(int n, const std::vector<float>& L, const std::vector<Var>& H, const std::vector<Var>& tau,
                        std::vector<Var>& x) {
    std::vector<float> b(n), xs(n, 0.f);
    for (int i = 0; i < n; ++i) b[i] = tau[i].v;
    dense_subs(n, L.data(), b.data(), xs.data());
    x.resize(n);
    SolveBlock blk;
    blk.n = n;
    blk.L = L;
    blk.H.resize(n * n);
    blk.tau.resize(n);
    blk.x.resize(n);
    blk.xv = xs;
    for (int i = 0; i < n * n; ++i) blk.H[i] = H[i].i;
    for (int i = 0; i < n; ++i) blk.tau[i] = tau[i].i;
    for (int i = 0; i < n; ++i) {
        x[i] = leaf(xs[i]);
        blk.x[i] = x[i].i;
    }
    blk.end = (int)g_tape->nodes.size();
    g_tape->blocks.push_back(std::move(blk));
}

void tape_reverse(std::vector<float>& adj) {
    Tape& t = *g_tape;
    int bi = (int)t.blocks.size() - 1;
    for (int i = (int)t.nodes.size() - 1; i >= 0; --i) {
        while (bi >= 0 && t.blocks[bi].end == i + 1) {
            SolveBlock& b = t.blocks[bi];
            int n = b.n;
            std::vector<float> ax(n), tmp(n, 0.f);
            for (int k = 0; k < n; ++k) ax[k] = adj[b.x[k]];
            dense_subs(n, b.L.data(), ax.data(), tmp.data());
            for (int k = 0; k < n; ++k)
                if (b.tau[k] >= 0) adj[b.tau[k]] += tmp[k];
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c)
                    if (b.H[r * n + c] >= 0) adj[b.H[r * n + c]] += -tmp[r] * b.xv[c];
            --bi;
        }
        const Node& nd = t.nodes[i];
        float a = adj[i];
        if (a == 0.0f) continue;
        for (int k = 0; k < 4; ++k)
            if (nd.p[k] >= 0) adj[nd.p[k]] += nd.d[k] * a;
    }
}

// ------------------------------------------------------------------------------------------------
template <class T> struct MassState {
    std::vector<T> H;      // n_qd*n_qd, (stale between updates)
    std::vector<float> L;  // Cholesky factor values
};

struct Debug {  // first-substep intermediates (float), all optional
    float *X_sc, *X_sm, *S_s, *I_s, *v_s, *a_s, *f_s, *ft_s, *tau, *qdd, *H, *L;
};

template <class T> inline void store_sv(float* dst, SV<T> s) {
    for (int k = 0; k < 6; ++k) dst[k] = val(s.get(k));
}
template <class T> inline void store_xf(float* dst, Xf<T> t) {
    dst[0] = val(t.p.x); dst[1] = val(t.p.y); dst[2] = val(t.p.z);
    dst[3] = val(t.q.x); dst[4] = val(t.q.y); dst[5] = val(t.q.z); dst[6] = val(t.q.w);
}

template <class T>
void substep(const dsim_model_desc& m, const T* q, const T* qd, const T* act, const T* mact, float dt,
             bool update_mass, MassState<T>& ms, T* q_new, T* qd_new, Debug* dbg) {
    const int L = m.n_links, nd = m.n_qd;
    std::vector<Xf<T>> X_sc(L), X_sm(L);
    std::vector<SV<T>> S_s(nd), v_s(L), a_s(L), f_s(L), ft_s(L);
    std::vector<M6<T>> I_s(L);
    std::vector<T> tau(nd, T(0.f));

    // ---- eval_rigid_fk
    for (int i = 0; i < L; ++i) {
        int parent = m.joint_parent[i];
        Xf<T> X_sp = xf_identity<T>();
        if (parent >= 0) X_sp = X_sc[parent];
        int type = m.joint_type[i];
        V3<T> axis = lift_v3<T>(m.joint_axis + 3 * i);
        int cs = m.joint_q_start[i];
        Xf<T> X_jc = xf_identity<T>();
        if (type == DSIM_JOINT_PRISMATIC) {
            X_jc.p = axis * q[cs];
        } else if (type == DSIM_JOINT_REVOLUTE) {
            X_jc.q = quat_from_axis_angle(axis, q[cs]);
        } else if (type == DSIM_JOINT_BALL) {
            X_jc.q = Q4<T>(q[cs + 0], q[cs + 1], q[cs + 2], q[cs + 3]);
        } else if (type == DSIM_JOINT_FREE) {
            X_jc.p = V3<T>(q[cs + 0], q[cs + 1], q[cs + 2]);
            X_jc.q = Q4<T>(q[cs + 3], q[cs + 4], q[cs + 5], q[cs + 6]);
        }
        Xf<T> X_pj = lift_xf<T>(load_xf(m.joint_X_pj + 7 * i));
        X_sc[i] = xf_mul(X_sp, xf_mul(X_pj, X_jc));
        Xf<T> X_cm = lift_xf<T>(load_xf(m.joint_X_cm + 7 * i));
        X_sm[i] = xf_mul(X_sc[i], X_cm);
    }

    // ---- eval_rigid_id
    V3<T> g = lift_v3<T>(m.gravity);
    for (int i = 0; i < L; ++i) {
        int type = m.joint_type[i];
        V3<T> axis = lift_v3<T>(m.joint_axis + 3 * i);
        int parent = m.joint_parent[i];
        int ds = m.joint_qd_start[i];
        Xf<T> X_sp = xf_identity<T>();
        if (parent >= 0) X_sp = X_sc[parent];
        Xf<T> X_pj = lift_xf<T>(load_xf(m.joint_X_pj + 7 * i));
        Xf<T> X_sj = xf_mul(X_sp, X_pj);
        SV<T> v_j;
        V3<T> zero3;
        if (type == DSIM_JOINT_PRISMATIC) {
            SV<T> S = transform_twist(X_sj, SV<T>(zero3, axis));
            v_j = S * qd[ds];
            S_s[ds] = S;
        } else if (type == DSIM_JOINT_REVOLUTE) {
            SV<T> S = transform_twist(X_sj, SV<T>(axis, zero3));
            v_j = S * qd[ds];
            S_s[ds] = S;
        } else if (type == DSIM_JOINT_BALL) {
            SV<T> S0 = transform_twist(X_sj, SV<T>(V3<T>(T(1.f), T(0.f), T(0.f)), zero3));
            SV<T> S1 = transform_twist(X_sj, SV<T>(V3<T>(T(0.f), T(1.f), T(0.f)), zero3));
            SV<T> S2 = transform_twist(X_sj, SV<T>(V3<T>(T(0.f), T(0.f), T(1.f)), zero3));
            S_s[ds + 0] = S0;
            S_s[ds + 1] = S1;
            S_s[ds + 2] = S2;
            v_j = S0 * qd[ds + 0] + S1 * qd[ds + 1] + S2 * qd[ds + 2];
        } else if (type == DSIM_JOINT_FREE) {
            v_j = SV<T>(V3<T>(qd[ds + 0], qd[ds + 1], qd[ds + 2]), V3<T>(qd[ds + 3], qd[ds + 4], qd[ds + 5]));
            for (int k = 0; k < 6; ++k) {
                SV<T> e;
                e.set(k, T(1.f));
                S_s[ds + k] = e;
            }
        }
        SV<T> v_parent, a_parent;
        if (parent >= 0) {
            v_parent = v_s[parent];
            a_parent = a_s[parent];
        }
        SV<T> v = v_parent + v_j;
        SV<T> a = a_parent + spatial_cross(v, v_j);
        M6<T> I_m;
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) I_m.d[r][c] = T(m.body_I_m[36 * i + 6 * r + c]);
        T mass = I_m.d[3][3];
        SV<T> f_g_m = SV<T>(V3<T>(), g) * mass;
        Xf<T> X_g;
        X_g.p = X_sm[i].p;
        X_g.q = quat_identity<T>();
        SV<T> f_g_s = transform_wrench(X_g, f_g_m);
        M6<T> Is = transform_inertia(X_sm[i], I_m);
        SV<T> f_b = m6vec(Is, a) + spatial_cross_dual(v, m6vec(Is, v));
        v_s[i] = v;
        a_s[i] = a;
        f_s[i] = f_b - f_g_s;
        I_s[i] = Is;
    }

    // ---- eval_rigid_contacts_art
    for (int c = 0; c < m.n_contacts; ++c) {
        int b = m.contact_body[c];
        V3<T> cp = lift_v3<T>(m.contact_point + 3 * c);
        T cd(m.contact_dist[c]);
        T ke(m.contact_material[4 * c + 0]), kd(m.contact_material[4 * c + 1]), kf(m.contact_material[4 * c + 2]),
            mu(m.contact_material[4 * c + 3]);
        V3<T> n(T(0.f), T(1.f), T(0.f));
        V3<T> p = xf_point(X_sc[b], cp) - n * cd;
        V3<T> w = v_s[b].w, v = v_s[b].v;
        V3<T> dpdt = v + cross(w, p);
        T cc = dot(n, p);
        if (val(cc) >= 0.0f) continue;
        T vn = dot(n, dpdt);
        V3<T> vt = dpdt - n * vn;
        T fn = cc * ke;
        T fd = s_min(vn, T(0.0f)) * kd * T(s_step(cc)) * (T(0.0f) - cc);
        V3<T> ft = normalize(vt) * s_min(kf * length(vt), T(0.0f) - mu * cc * ke) * T(s_step(cc));
        V3<T> f_total = n * (fn + fd) + ft;
        V3<T> t_total = cross(p, f_total);
        f_s[b] = f_s[b] + SV<T>(t_total, f_total);
    }

    // ---- eval_muscles
    for (int mi = 0; mi < m.n_muscles; ++mi) {
        int s = m.muscle_start[mi], e = m.muscle_start[mi + 1] - 1;
        T activation = mact[mi];
        for (int i = s; i < e; ++i) {
            int l0 = m.muscle_links[i], l1 = m.muscle_links[i + 1];
            if (l0 == l1) continue;
            V3<T> r0 = lift_v3<T>(m.muscle_points + 3 * i), r1 = lift_v3<T>(m.muscle_points + 3 * (i + 1));
            V3<T> pos0 = xf_point(X_sc[l0], r0), pos1 = xf_point(X_sc[l1], r1);
            V3<T> n = normalize(pos1 - pos0);
            V3<T> f = n * activation;
            f_s[l0] = f_s[l0] - SV<T>(cross(pos0, f), f);
            f_s[l1] = f_s[l1] + SV<T>(cross(pos1, f), f);
        }
    }

    // ---- eval_rigid_tau (reverse link order)
    for (int off = 0; off < L; ++off) {
        int i = L - off - 1;
        int type = m.joint_type[i], parent = m.joint_parent[i];
        int ds = m.joint_qd_start[i], cs = m.joint_q_start[i];
        T tke(m.joint_target_ke[i]), tkd(m.joint_target_kd[i]), lke(m.joint_limit_ke[i]), lkd(m.joint_limit_kd[i]);
        SV<T> f = f_s[i] + ft_s[i];
        if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
            T qq = q[cs], qdv = qd[ds], a = act[ds];
            T target(m.joint_target[cs]), lower(m.joint_limit_lower[cs]), upper(m.joint_limit_upper[cs]);
            T limit_f(0.0f);
            if (val(qq) < val(lower)) limit_f = lke * (lower - qq);
            if (val(qq) > val(upper)) limit_f = lke * (upper - qq);
            T damping_f = (T(0.0f) - lkd) * qdv;
            T t = T(0.0f) - spatial_dot(S_s[ds], f) - tke * (qq - target) - tkd * qdv + a + limit_f + damping_f;
            tau[ds] = t;
        } else if (type == DSIM_JOINT_BALL) {
            for (int k = 0; k < 3; ++k)
                tau[ds + k] = T(0.0f) - spatial_dot(S_s[ds + k], f) - qd[ds + k] * tkd - q[cs + k] * tke;
        } else if (type == DSIM_JOINT_FREE) {
            for (int k = 0; k < 6; ++k) tau[ds + k] = T(0.0f) - spatial_dot(S_s[ds + k], f);
        }
        if (parent >= 0) ft_s[parent] = ft_s[parent] + f;
    }

    // ---- mass matrix (update substeps only)
    if (update_mass) {
        const int R = 6 * L;
        std::vector<T> J((size_t)R * nd, T(0.f)), M((size_t)R * R, T(0.f)), P((size_t)R * nd, T(0.f));
        for (int i = 0; i < L; ++i) {  // spatial_jacobian, spatial.h:691-738
            int j = i;
            while (j != -1) {
                for (int col = m.joint_qd_start[j]; col < m.joint_qd_start[j + 1]; ++col)
                    for (int r = 0; r < 6; ++r) J[(size_t)(6 * i + r) * nd + col] = S_s[col].get(r);
                j = m.joint_parent[j];
            }
        }
        for (int l = 0; l < L; ++l)  // spatial_mass, spatial.h:801-815
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) M[(size_t)(6 * l + r) * R + 6 * l + c] = I_s[l].d[r][c];
        for (int i = 0; i < R; ++i)  // P = M*J, dense_gemm_impl<false,false>
            for (int j = 0; j < nd; ++j) {
                T s(0.f);
                for (int k = 0; k < R; ++k) s += M[(size_t)i * R + k] * J[(size_t)k * nd + j];
                P[(size_t)i * nd + j] = s;
            }
        ms.H.assign((size_t)nd * nd, T(0.f));
        for (int i = 0; i < nd; ++i)  // H = J^T*P, dense_gemm_impl<true,false>
            for (int j = 0; j < nd; ++j) {
                T s(0.f);
                for (int k = 0; k < R; ++k) s += J[(size_t)k * nd + i] * P[(size_t)k * nd + j];
                ms.H[(size_t)i * nd + j] = s;
            }
        std::vector<float> Hv((size_t)nd * nd);
        for (size_t k = 0; k < Hv.size(); ++k) Hv[k] = val(ms.H[k]);
        ms.L.assign((size_t)nd * nd, 0.f);
        dense_chol(nd, Hv.data(), m.joint_armature, ms.L.data());
    }

    // ---- solve
    std::vector<T> qdd;
    dense_solve(nd, ms.L, ms.H, tau, qdd);

    // ---- eval_rigid_integrate
    T h(dt);
    for (int i = 0; i < L; ++i) {
        int type = m.joint_type[i], cs = m.joint_q_start[i], ds = m.joint_qd_start[i];
        if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
            T qd_n = qd[ds] + qdd[ds] * h;
            T q_n = q[cs] + qd_n * h;
            qd_new[ds] = qd_n;
            q_new[cs] = q_n;
        } else if (type == DSIM_JOINT_BALL) {
            V3<T> mj(qdd[ds], qdd[ds + 1], qdd[ds + 2]), wj(qd[ds], qd[ds + 1], qd[ds + 2]);
            Q4<T> r(q[cs], q[cs + 1], q[cs + 2], q[cs + 3]);
            V3<T> wn = wj + mj * h;
            Q4<T> dr = qmul(Q4<T>(wn.x, wn.y, wn.z, T(0.0f)), r);
            dr = Q4<T>(dr.x * T(0.5f), dr.y * T(0.5f), dr.z * T(0.5f), dr.w * T(0.5f));
            Q4<T> rn = qnormalize(Q4<T>(r.x + dr.x * h, r.y + dr.y * h, r.z + dr.z * h, r.w + dr.w * h));
            q_new[cs] = rn.x; q_new[cs + 1] = rn.y; q_new[cs + 2] = rn.z; q_new[cs + 3] = rn.w;
            qd_new[ds] = wn.x; qd_new[ds + 1] = wn.y; qd_new[ds + 2] = wn.z;
        } else if (type == DSIM_JOINT_FREE) {
            V3<T> ms_(qdd[ds], qdd[ds + 1], qdd[ds + 2]), as_(qdd[ds + 3], qdd[ds + 4], qdd[ds + 5]);
            V3<T> w(qd[ds], qd[ds + 1], qd[ds + 2]), v(qd[ds + 3], qd[ds + 4], qd[ds + 5]);
            w = w + ms_ * h;
            v = v + as_ * h;
            V3<T> p(q[cs], q[cs + 1], q[cs + 2]);
            V3<T> dpdt = v + cross(w, p);
            Q4<T> r(q[cs + 3], q[cs + 4], q[cs + 5], q[cs + 6]);
            Q4<T> dr = qmul(Q4<T>(w.x, w.y, w.z, T(0.0f)), r);
            dr = Q4<T>(dr.x * T(0.5f), dr.y * T(0.5f), dr.z * T(0.5f), dr.w * T(0.5f));
            V3<T> pn = p + dpdt * h;
            Q4<T> rn = qnormalize(Q4<T>(r.x + dr.x * h, r.y + dr.y * h, r.z + dr.z * h, r.w + dr.w * h));
            q_new[cs] = pn.x; q_new[cs + 1] = pn.y; q_new[cs + 2] = pn.z;
            q_new[cs + 3] = rn.x; q_new[cs + 4] = rn.y; q_new[cs + 5] = rn.z; q_new[cs + 6] = rn.w;
            qd_new[ds] = w.x; qd_new[ds + 1] = w.y; qd_new[ds + 2] = w.z;
            qd_new[ds + 3] = v.x; qd_new[ds + 4] = v.y; qd_new[ds + 5] = v.z;
        }
    }

    if (dbg) {
        for (int i = 0; i < L; ++i) {
            if (dbg->X_sc) store_xf(dbg->X_sc + 7 * i, X_sc[i]);
            if (dbg->X_sm) store_xf(dbg->X_sm + 7 * i, X_sm[i]);
            if (dbg->v_s) store_sv(dbg->v_s + 6 * i, v_s[i]);
            if (dbg->a_s) store_sv(dbg->a_s + 6 * i, a_s[i]);
            if (dbg->f_s) store_sv(dbg->f_s + 6 * i, f_s[i]);
            if (dbg->ft_s) store_sv(dbg->ft_s + 6 * i, ft_s[i]);
            if (dbg->I_s)
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) dbg->I_s[36 * i + 6 * r + c] = val(I_s[i].d[r][c]);
        }
        for (int k = 0; k < nd; ++k) {
            if (dbg->S_s) store_sv(dbg->S_s + 6 * k, S_s[k]);
            if (dbg->tau) dbg->tau[k] = val(tau[k]);
            if (dbg->qdd) dbg->qdd[k] = val(qdd[k]);
        }
        for (int k = 0; k < nd * nd; ++k) {
            if (dbg->H) dbg->H[k] = val(ms.H[k]);
            if (dbg->L) dbg->L[k] = ms.L[k];
        }
    }
}

template <class T>
void env_step(const dsim_model_desc& m, std::vector<T>& q, std::vector<T>& qd, const std::vector<T>& act,
              const std::vector<T>& mact, float dt, int substeps, int mm_freq, Debug* dbg) {
    MassState<T> ms;
    std::vector<T> qn(m.n_q, T(0.f)), qdn(m.n_qd, T(0.f));
    const float h = dt / float(substeps);
    for (int s = 0; s < substeps; ++s) {
        substep(m, q.data(), qd.data(), act.data(), mact.data(), h, (s % mm_freq) == 0, ms, qn.data(), qdn.data(),
                s == 0 ? dbg : nullptr);
        q = qn;
        qd = qdn;
    }
}

}  // namespace

extern "C" {

// Forward only, fp32 (plain float path).  dbg_* may be NULL; they receive the first substep's
// intermediates of every env ([N][...] env-major).
int dsim_oracle_step_forward(const dsim_model_desc* m, int n_envs, const float* q_in, const float* qd_in,
                             const float* act, const float* muscle_act, float dt, int substeps, int mm_freq,
                             float* q_out, float* qd_out, float* dbg_X_sc, float* dbg_X_sm, float* dbg_S_s,
                             float* dbg_I_s, float* dbg_v_s, float* dbg_a_s, float* dbg_f_s, float* dbg_ft_s,
                             float* dbg_tau, float* dbg_qdd, float* dbg_H, float* dbg_L) {
    const int nq = m->n_q, nd = m->n_qd, M = m->n_muscles, L = m->n_links;
    for (int e = 0; e < n_envs; ++e) {
        std::vector<float> q(q_in + (size_t)e * nq, q_in + (size_t)(e + 1) * nq);
        std::vector<float> qd(qd_in + (size_t)e * nd, qd_in + (size_t)(e + 1) * nd);
        std::vector<float> a(act + (size_t)e * nd, act + (size_t)(e + 1) * nd);
        std::vector<float> ma;
        if (M > 0) ma.assign(muscle_act + (size_t)e * M, muscle_act + (size_t)(e + 1) * M);
        Debug d;
        d.X_sc = dbg_X_sc ? dbg_X_sc + (size_t)e * L * 7 : nullptr;
        d.X_sm = dbg_X_sm ? dbg_X_sm + (size_t)e * L * 7 : nullptr;
        d.S_s = dbg_S_s ? dbg_S_s + (size_t)e * nd * 6 : nullptr;
        d.I_s = dbg_I_s ? dbg_I_s + (size_t)e * L * 36 : nullptr;
        d.v_s = dbg_v_s ? dbg_v_s + (size_t)e * L * 6 : nullptr;
        d.a_s = dbg_a_s ? dbg_a_s + (size_t)e * L * 6 : nullptr;
        d.f_s = dbg_f_s ? dbg_f_s + (size_t)e * L * 6 : nullptr;
        d.ft_s = dbg_ft_s ? dbg_ft_s + (size_t)e * L * 6 : nullptr;
        d.tau = dbg_tau ? dbg_tau + (size_t)e * nd : nullptr;
        d.qdd = dbg_qdd ? dbg_qdd + (size_t)e * nd : nullptr;
        d.H = dbg_H ? dbg_H + (size_t)e * nd * nd : nullptr;
        d.L = dbg_L ? dbg_L + (size_t)e * nd * nd : nullptr;
        env_step<float>(*m, q, qd, a, ma, dt, substeps, mm_freq, &d);
        memcpy(q_out + (size_t)e * nq, q.data(), sizeof(float) * nq);
        memcpy(qd_out + (size_t)e * nd, qd.data(), sizeof(float) * nd);
    }
    return 0;
}

// Forward (taped) + reverse sweep of environments [e0, e1).  Gradients are written (not accumulated).
static void backward_range(const dsim_model_desc* m, int e0, int e1, const float* q_in, const float* qd_in,
                           const float* act, const float* muscle_act, float dt, int substeps, int mm_freq,
                           const float* gq_out, const float* gqd_out, float* gq_in, float* gqd_in, float* gact,
                           float* gmuscle_act, float* q_out, float* qd_out) {
    const int nq = m->n_q, nd = m->n_qd, M = m->n_muscles;
    Tape tape;
    g_tape = &tape;
    for (int e = e0; e < e1; ++e) {
        tape.clear();
        std::vector<Var> q(nq), qd(nd), a(nd), ma(M);
        for (int k = 0; k < nq; ++k) q[k] = leaf(q_in[(size_t)e * nq + k]);
        for (int k = 0; k < nd; ++k) qd[k] = leaf(qd_in[(size_t)e * nd + k]);
        for (int k = 0; k < nd; ++k) a[k] = leaf(act[(size_t)e * nd + k]);
        for (int k = 0; k < M; ++k) ma[k] = leaf(muscle_act[(size_t)e * M + k]);
        std::vector<Var> q0 = q, qd0 = qd;
        env_step<Var>(*m, q, qd, a, ma, dt, substeps, mm_freq, nullptr);
        std::vector<float> adj(tape.nodes.size(), 0.f);
        for (int k = 0; k < nq; ++k)
            if (q[k].i >= 0) adj[q[k].i] += gq_out[(size_t)e * nq + k];
        for (int k = 0; k < nd; ++k)
            if (qd[k].i >= 0) adj[qd[k].i] += gqd_out[(size_t)e * nd + k];
        tape_reverse(adj);
        for (int k = 0; k < nq; ++k) gq_in[(size_t)e * nq + k] = adj[q0[k].i];
        for (int k = 0; k < nd; ++k) gqd_in[(size_t)e * nd + k] = adj[qd0[k].i];
        if (gact)
            for (int k = 0; k < nd; ++k) gact[(size_t)e * nd + k] = adj[a[k].i];
        if (gmuscle_act)
            for (int k = 0; k < M; ++k) gmuscle_act[(size_t)e * M + k] = adj[ma[k].i];
        if (q_out)
            for (int k = 0; k < nq; ++k) q_out[(size_t)e * nq + k] = q[k].v;
        if (qd_out)
            for (int k = 0; k < nd; ++k) qd_out[(size_t)e * nd + k] = qd[k].v;
    }
    g_tape = nullptr;
}

int dsim_oracle_step_backward(const dsim_model_desc* m, int n_envs, const float* q_in, const float* qd_in,
                              const float* act, const float* muscle_act, float dt, int substeps, int mm_freq,
                              const float* gq_out, const float* gqd_out, float* gq_in, float* gqd_in, float* gact,
                              float* gmuscle_act, float* q_out, float* qd_out) {
    backward_range(m, 0, n_envs, q_in, qd_in, act, muscle_act, dt, substeps, mm_freq, gq_out, gqd_out, gq_in, gqd_in,
                   gact, gmuscle_act, q_out, qd_out);
    return 0;
}

// The same over `n_threads` host threads (environments are independent: contiguous env ranges, one tape per
// thread; results are identical to the single-threaded call).  Used by the CPU-baseline leg of bench.py, which
// reports the thread count it ran with.
int dsim_oracle_step_backward_mt(const dsim_model_desc* m, int n_envs, int n_threads, const float* q_in,
                                 const float* qd_in, const float* act, const float* muscle_act, float dt, int substeps,
                                 int mm_freq, const float* gq_out, const float* gqd_out, float* gq_in, float* gqd_in,
                                 float* gact, float* gmuscle_act, float* q_out, float* qd_out) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_envs) n_threads = n_envs;
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) {
        const int e0 = (int)((long long)n_envs * t / n_threads), e1 = (int)((long long)n_envs * (t + 1) / n_threads);
        pool.emplace_back(backward_range, m, e0, e1, q_in, qd_in, act, muscle_act, dt, substeps, mm_freq, gq_out, gqd_out,
                          gq_in, gqd_in, gact, gmuscle_act, q_out, qd_out);
    }
    for (auto& th : pool) th.join();
    return 0;
}

}  // extern "C"
