// dsim_math.hpp -- small fp32 vector/quaternion/spatial algebra for the gfx950 kernels.
//
// Hand-written equivalents of the reference's header math library (dflex/dflex/vec3.h, quat.h,
// spatial.h) restricted to what the articulated path needs, plus the reverse-mode rules used by the
// hand-derived adjoint.  Conventions are the reference's: quaternions are (x,y,z,w)
// (quat.h:8-20), spatial vectors are (angular, linear) (spatial.h:6-9).
//
// The including translation unit defines DSIM_FN (the .hip file: `__device__ __forceinline__`;
// the lane-serial unit-test harness: `static inline`).
#pragma once
#include <math.h>

struct v3 {
    float x, y, z;
};
struct q4 {
    float x, y, z, w;
};
struct sv6 {  // spatial vector: w = angular part, v = linear part
    v3 w, v;
};

DSIM_FN v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
DSIM_FN v3 zero3() { return v3{0.f, 0.f, 0.f}; }
DSIM_FN v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DSIM_FN v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DSIM_FN v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
DSIM_FN v3 operator*(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
DSIM_FN v3 operator*(float s, v3 a) { return v3{a.x * s, a.y * s, a.z * s}; }
DSIM_FN void operator+=(v3& a, v3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
DSIM_FN void operator-=(v3& a, v3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; }
DSIM_FN float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DSIM_FN v3 cross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
DSIM_FN v3 ld3(const float* p) { return v3{p[0], p[1], p[2]}; }
DSIM_FN void st3(float* p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
DSIM_FN void add3(float* p, v3 a) { p[0] += a.x; p[1] += a.y; p[2] += a.z; }

DSIM_FN q4 mkq(float x, float y, float z, float w) { return q4{x, y, z, w}; }
DSIM_FN q4 ldq(const float* p) { return q4{p[0], p[1], p[2], p[3]}; }
DSIM_FN void stq(float* p, q4 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; }
DSIM_FN void addq(float* p, q4 a) { p[0] += a.x; p[1] += a.y; p[2] += a.z; p[3] += a.w; }
DSIM_FN q4 operator+(q4 a, q4 b) { return q4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
DSIM_FN q4 operator*(q4 a, float s) { return q4{a.x * s, a.y * s, a.z * s, a.w * s}; }
DSIM_FN void operator+=(q4& a, q4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
DSIM_FN float qdot(q4 a, q4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
DSIM_FN q4 qconj(q4 a) { return q4{-a.x, -a.y, -a.z, a.w}; }
DSIM_FN v3 qvec(q4 a) { return v3{a.x, a.y, a.z}; }
// Hamilton product, quat.h:100-106
DSIM_FN q4 qmul(q4 a, q4 b) {
    return q4{a.w * b.x + b.w * a.x + a.y * b.z - b.y * a.z,
              a.w * b.y + b.w * a.y + a.z * b.x - b.z * a.x,
              a.w * b.z + b.w * a.z + a.x * b.y - b.x * a.y,
              a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
// (a, 0) (x) b for a pure-vector left factor (angular velocity, torque): qmul's formulas without the four products with the
// literal 0 -- which IEEE arithmetic does not let the compiler drop, although 0 * x + y is y for every finite x
DSIM_FN q4 qmul_v(v3 a, q4 b) {
    return q4{b.w * a.x + a.y * b.z - b.y * a.z,
              b.w * a.y + a.z * b.x - b.z * a.x,
              b.w * a.z + a.x * b.y - b.x * a.y,
              -(a.x * b.x) - a.y * b.y - a.z * b.z};
}
// c = a (x) b is bilinear: adj_a = adj_c (x) conj(b), adj_b = conj(a) (x) adj_c  (== quat.h:232-247)
DSIM_FN q4 qmul_adj_a(q4 b, q4 r) { return qmul(r, qconj(b)); }
DSIM_FN q4 qmul_adj_b(q4 a, q4 r) { return qmul(qconj(a), r); }

// rotate(q, x) = x(2w^2-1) + 2w (qv x x) + 2 qv (qv.x), quat.h:113-116
DSIM_FN v3 rotate(q4 q, v3 x) {
    v3 qv = qvec(q);
    return x * (2.0f * q.w * q.w - 1.0f) + cross(qv, x) * (q.w * 2.0f) + qv * (dot(qv, x) * 2.0f);
}
// columns of the rotation matrix: rotate(q, e_x), rotate(q, e_y), rotate(q, e_z) with the exact-zero terms of the
// formula above left out (0 * finite and + 0 are exact, so the values are those of three rotate() calls)
DSIM_FN void rotate_basis(q4 q, v3& rx, v3& ry, v3& rz) {
    const float a = 2.0f * q.w * q.w - 1.0f, b = q.w * 2.0f;
    const float cx = q.x * 2.0f, cy = q.y * 2.0f, cz = q.z * 2.0f;
    rx = v3{a + q.x * cx, q.z * b + q.y * cx, -q.y * b + q.z * cx};
    ry = v3{-q.z * b + q.x * cy, a + q.y * cy, q.x * b + q.z * cy};
    rz = v3{q.y * b + q.x * cz, -q.x * b + q.y * cz, a + q.z * cz};
}
// R(q)^T r : adjoint of rotate w.r.t. x
DSIM_FN v3 rotate_inv(q4 q, v3 r) {
    v3 qv = qvec(q);
    return r * (2.0f * q.w * q.w - 1.0f) - cross(qv, r) * (q.w * 2.0f) + qv * (dot(qv, r) * 2.0f);
}
// adjoint of rotate w.r.t. q for cotangent r (same function as quat.h:256-288, derived by hand):
//   d/dw : 4w (x.r) + 2 (qv x x).r ;  d/dqv : 2w (x x r) + 2 (qv.r) x + 2 (qv.x) r
DSIM_FN q4 rotate_adj_q(q4 q, v3 x, v3 r) {
    v3 qv = qvec(q);
    float aw = 4.0f * q.w * dot(x, r) + 2.0f * dot(cross(qv, x), r);
    v3 av = cross(x, r) * (2.0f * q.w) + x * (2.0f * dot(qv, r)) + r * (2.0f * dot(qv, x));
    return q4{av.x, av.y, av.z, aw};
}
// 1 / sqrt(n2) for the normalisation of a quaternion (0 for n2 == 0: the reference's normalize() leaves a zero quaternion alone,
// quat.h:70-83).  Device code: v_rsq_f32 + one Newton step -- as accurate as the reference's two correctly rounded operations
// (square root, division) together (< 1 ulp), ~6 instead of ~30 dependent instructions on the one lane that integrates the free
// root; not the same bits (-DDSIM_EXACT_DIV_SQRT builds the A/B variant; the host harness of tests/emu always takes that path).
// v_rsq_f32 flushes a denormal input to zero: rsq = +inf, and the correction steps then make inf - inf = NaN.  A squared length
// below the smallest normal (a tangential contact velocity of ~1e-19, a quaternion of that norm) is treated as zero length,
// like the exactly-zero case: the force / rotation it would scale is below 1e-15 of anything else in the step.
#define DSIM_MIN_NORMAL 1.17549435e-38f
DSIM_FN float dsim_inv_len_exact(float n2) {
    const float l = sqrtf(n2);
    return l > 0.0f ? 1.0f / l : 0.0f;
}
DSIM_FN float dsim_inv_len_fast(float n2) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(n2);
    const float r = y * __builtin_fmaf(-0.5f * n2 * y, y, 1.5f);
    return n2 >= DSIM_MIN_NORMAL ? r : 0.0f;
#else
    return dsim_inv_len_exact(n2);
#endif
}
// The reference's two operations -- l = sqrt(n2), then 1 / l -- each from the hardware approximation plus one residual correction
// (sqrt: l = n2 rsq(n2), l += (n2 - l l) rsq(n2) / 2; reciprocal: c = rcp(l), c += (1 - l c) c): the corrected values are the
// correctly rounded ones except for rare halfway cases, so the result follows the reference's roundings instead of replacing two
// roundings by a differently rounded 1 / sqrt -- which matters where it is applied 512 times in a row to the same quantity.
DSIM_FN float dsim_inv_len_two_step(float n2) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(n2);
    float l = n2 * y;
    l = __builtin_fmaf(__builtin_fmaf(-l, l, n2), 0.5f * y, l);
    float c = __builtin_amdgcn_rcpf(l);
    c = __builtin_fmaf(__builtin_fmaf(-l, c, 1.0f), c, c);
    return n2 >= DSIM_MIN_NORMAL ? c : 0.0f;
#else
    return dsim_inv_len_exact(n2);
#endif
}
// ... of a quaternion in the integrator (one lane, on the critical path of every substep) and of a contact's tangential velocity.
// Round 5 measured the plain v_rsq_f32 + Newton form at these two sites against the reference's recordings: in the integrator
// 15 instead of 3 of the 512 sampled Ant environments left the 1e-3 band, in the contact friction 14 instead of 8 of the 128
// sampled Humanoid environments (tools/ant_grad_probe.py; the pivots' v_rcp_f32 + Newton and the muscle segments' one-step
// form changed nothing) -- these are the values friction regimes and contact thresholds hang on, and the quantity the integrator
// renormalises 512 times per rollout.  They keep the reference's two roundings (dsim_inv_len_two_step: same counts as the
// correctly rounded operations; -DDSIM_INTEG_RSQ builds the one-step form).
DSIM_FN float dsim_inv_len(float n2) {
#if defined(DSIM_EXACT_DIV_SQRT) || defined(DSIM_EXACT_RSQ_INTEG)
    return dsim_inv_len_exact(n2);
#elif defined(DSIM_INTEG_RSQ)
    return dsim_inv_len_fast(n2);
#else
    return dsim_inv_len_two_step(n2);
#endif
}
// ... of a muscle segment (per-item lanes; a smooth function of the poses, no thresholds)
DSIM_FN float dsim_inv_len_item(float n2) {
#if defined(DSIM_EXACT_DIV_SQRT) || defined(DSIM_EXACT_RSQ_ITEM)
    return dsim_inv_len_exact(n2);
#else
    return dsim_inv_len_fast(n2);
#endif
}
// sin/cos of a joint half-angle.  Joint angles live inside their limits (|q| <= ~pi), so |x| <= pi/2 is the
// hot case: odd/even Taylor polynomials to x^11 / x^12 (truncation error < 6e-8 at pi/2, i.e. below fp32
// resolution) -- ~14 FMAs instead of two library calls with full range reduction; anything larger falls
// back to the precise library routines.
DSIM_FN void half_angle_sincos(float x, float& s, float& c) {
    if (fabsf(x) <= 1.5707964f) {
        const float z = x * x;
        float ps = -2.5052108e-08f;
        ps = ps * z + 2.7557319e-06f;
        ps = ps * z - 1.9841270e-04f;
        ps = ps * z + 8.3333333e-03f;
        ps = ps * z - 1.6666667e-01f;
        s = x + x * z * ps;
        float pc = 2.0876757e-09f;
        pc = pc * z - 2.7557319e-07f;
        pc = pc * z + 2.4801587e-05f;
        pc = pc * z - 1.3888889e-03f;
        pc = pc * z + 4.1666667e-02f;
        pc = pc * z - 0.5f;
        c = 1.0f + z * pc;
    } else {
        s = sinf(x);
        c = cosf(x);
    }
}
// quat.h:44-52
DSIM_FN q4 quat_axis_angle(v3 axis, float angle) {
    float s, c;
    half_angle_sincos(angle * 0.5f, s, c);
    return q4{axis.x * s, axis.y * s, axis.z * s, c};
}
// d/d angle of the above dotted with cotangent r (quat.h:153-164)
DSIM_FN float quat_axis_angle_adj(v3 axis, float angle, q4 r) {
    float s, c;
    half_angle_sincos(angle * 0.5f, s, c);
    return 0.5f * (c * (axis.x * r.x + axis.y * r.y + axis.z * r.z) - s * r.w);
}

DSIM_FN sv6 mksv(v3 w, v3 v) { return sv6{w, v}; }
DSIM_FN sv6 zerosv() { return sv6{zero3(), zero3()}; }
DSIM_FN sv6 operator+(sv6 a, sv6 b) { return sv6{a.w + b.w, a.v + b.v}; }
DSIM_FN sv6 operator-(sv6 a, sv6 b) { return sv6{a.w - b.w, a.v - b.v}; }
DSIM_FN sv6 operator*(sv6 a, float s) { return sv6{a.w * s, a.v * s}; }
DSIM_FN void operator+=(sv6& a, sv6 b) { a.w += b.w; a.v += b.v; }
DSIM_FN float sdot(sv6 a, sv6 b) { return dot(a.w, b.w) + dot(a.v, b.v); }
DSIM_FN sv6 ldsv(const float* p) { return sv6{ld3(p), ld3(p + 3)}; }
DSIM_FN void stsv(float* p, sv6 a) { st3(p, a.w); st3(p + 3, a.v); }
// spatial.h:56-70
DSIM_FN sv6 scross(sv6 a, sv6 b) { return sv6{cross(a.w, b.w), cross(a.v, b.w) + cross(a.w, b.v)}; }
DSIM_FN sv6 scross_dual(sv6 a, sv6 b) { return sv6{cross(a.w, b.w) + cross(a.v, b.v), cross(a.w, b.v)}; }

// Rigid-body spatial inertia about the world origin in 10 parameters:
//   I = [[A, [h]x], [-[h]x, m 1]],  A symmetric (xx,xy,xz,yy,yz,zz), h = m c (first moment)
// This is what the reference's dense 6x6 `T^T I_m T` (sim.py:1117-1134) evaluates to for
// I_m = diag(Ic, m 1); sums of such matrices (composite bodies) keep the form.
struct inertia10 {
    float m;
    v3 h;
    float axx, axy, axz, ayy, ayz, azz;
};
DSIM_FN v3 sym_mul(const inertia10& I, v3 x) {
    return v3{I.axx * x.x + I.axy * x.y + I.axz * x.z, I.axy * x.x + I.ayy * x.y + I.ayz * x.z,
              I.axz * x.x + I.ayz * x.y + I.azz * x.z};
}
DSIM_FN sv6 inertia_mul(const inertia10& I, sv6 x) {
    return sv6{sym_mul(I, x.w) + cross(I.h, x.v), x.v * I.m + cross(x.w, I.h)};
}
DSIM_FN inertia10 ld_i10(const float* p) {
    inertia10 I;
    I.m = p[0]; I.h = ld3(p + 1);
    I.axx = p[4]; I.axy = p[5]; I.axz = p[6]; I.ayy = p[7]; I.ayz = p[8]; I.azz = p[9];
    return I;
}
DSIM_FN void st_i10(float* p, const inertia10& I) {
    p[0] = I.m; st3(p + 1, I.h);
    p[4] = I.axx; p[5] = I.axy; p[6] = I.axz; p[7] = I.ayy; p[8] = I.ayz; p[9] = I.azz;
}
// cotangent of y^T I x w.r.t. the 10 parameters, scaled by wgt and accumulated into g[10]
// (g[0] (mass) is never needed: mass is constant)
DSIM_FN void inertia_bilinear_adj(float* g, sv6 y, sv6 x, float wgt) {
    v3 gh = (cross(x.v, y.w) + cross(y.v, x.w)) * wgt;
    g[1] += gh.x; g[2] += gh.y; g[3] += gh.z;
    g[4] += wgt * (y.w.x * x.w.x);
    g[5] += wgt * (y.w.x * x.w.y + y.w.y * x.w.x);
    g[6] += wgt * (y.w.x * x.w.z + y.w.z * x.w.x);
    g[7] += wgt * (y.w.y * x.w.y);
    g[8] += wgt * (y.w.y * x.w.z + y.w.z * x.w.y);
    g[9] += wgt * (y.w.z * x.w.z);
}
