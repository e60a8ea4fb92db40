// dsim_literal.hpp -- the RADIAL component of dL/d joint_q[quaternion block] as the reference's literal formulas give it
// (opt-in: dsim_step_backward_literal, include/dsim.h).
//
// The adjoint kernels keep pose cotangents as world-frame wrenches; a wrench has no component along the direction in which
// a quaternion's NORM changes, so for every quaternion block of joint_q they return the reference's cotangent minus its
// component along the quaternion.  The reference differentiates its formulas literally
// (dflex/dflex/quat.h:232-288 adj_mul / adj_rotate, spatial.h:740-798), also off the unit sphere, where rotate() is a
// rotation times |q|^2 (quat.h:113-116) and T^T I_m T is built from a non-orthogonal R (sim.py:1117-1134).  That component is
//     rho_j = dL / d eps_j   under   q_j -> (1 + eps_j) q_j     (q_j: the quaternion block of joint j, |q_j| = 1),
// and the literal cotangent is the kernels' value + rho_j q_j.  rho_j is one directional derivative, so it is evaluated in
// FORWARD mode: the first substep (the only one that sees joint_q as handed in -- every later one starts from quaternions the
// integrator normalised, sim.py:1552, 1616) is re-run on dual numbers with the reference's literal formulas in the reference's
// own form, and its tangent is contracted with what the adjoint launch already holds at that point:
//     rho_j = < gq_1, d q_1 > + < gqd_1, d qd_1 > + < adj_H, d H >
// gq_1 / gqd_1: cotangents of the first substep's outputs; adj_H: the cotangent of the mass matrix accumulated over the
// substeps that reuse the first factor (matnn.h:310-336: -t qdd^T per substep; the Cholesky adjoint is a no-op, so H enters
// through the solves only); d qdd = H^-1 d tau with H held fixed.  The conventions of the reference's reverse mode that are not
// plain calculus hold here in forward mode by the same branches: min -> the selected argument, ties to the second
// (adjoint.h:129-143), step() constant (177-180), length / normalize of a zero vector constant (vec3.h:194-222), a
// non-penetrating contact contributes nothing (sim.py:1179-1180).
//
// Formulas restated from: sim.py:1269-1319 (jcalc_transform), 1638-1678 (FK), 1076-1134 (twist / wrench / inertia transforms),
// 1323-1387 (jcalc_motion), 1716-1789 (ID), 1137-1206 (contacts), 1209-1265 (muscles), 1421-1502 + 1792-1842 (tau),
// spatial.h:691-815 + matnn.h:23-99 (H = J^T M J, evaluated link by link: the same polynomial in S and I_s), 1505-1636
// (integrate), quat.h:44-121, spatial.h:166-357.  One thread per (environment, quaternion joint); plain sequential code, local
// arrays (scratch memory): this is a correctness path of one substep, not a hot kernel.
#pragma once
#include "dsim_layout.hpp"

#ifndef DSIM_LIT_FN
#define DSIM_LIT_FN inline
#endif

#define DSIM_LIT_LMAX 24    // links
#define DSIM_LIT_NDMAX 28   // dofs

namespace dsim_lit {

struct Du {   // value and derivative with respect to eps
    float v, d;
    DSIM_LIT_FN Du() : v(0.f), d(0.f) {}
    DSIM_LIT_FN Du(float a) : v(a), d(0.f) {}
    DSIM_LIT_FN Du(float a, float b) : v(a), d(b) {}
};
DSIM_LIT_FN Du operator+(Du a, Du b) { return Du(a.v + b.v, a.d + b.d); }
DSIM_LIT_FN Du operator-(Du a, Du b) { return Du(a.v - b.v, a.d - b.d); }
DSIM_LIT_FN Du operator-(Du a) { return Du(-a.v, -a.d); }
DSIM_LIT_FN Du operator*(Du a, Du b) { return Du(a.v * b.v, a.d * b.v + a.v * b.d); }
DSIM_LIT_FN Du operator/(Du a, Du b) {
    const float r = a.v / b.v;
    return Du(r, (a.d - r * b.d) / b.v);
}
DSIM_LIT_FN Du dsqrt(Du a) {
    const float r = sqrtf(a.v);
    return Du(r, r > 0.f ? 0.5f * a.d / r : 0.f);
}
DSIM_LIT_FN Du dmin(Du a, Du b) { return a.v < b.v ? a : b; }   // adjoint.h:129-143: ties to the second argument
DSIM_LIT_FN float dstep(Du x) { return x.v < 0.0f ? 1.0f : 0.0f; }   // adjoint.h:99, 177-180

struct V3 {
    Du x, y, z;
    DSIM_LIT_FN V3() {}
    DSIM_LIT_FN V3(Du a, Du b, Du c) : x(a), y(b), z(c) {}
};
DSIM_LIT_FN V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
DSIM_LIT_FN V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
DSIM_LIT_FN V3 operator*(V3 a, Du s) { return V3(a.x * s, a.y * s, a.z * s); }
DSIM_LIT_FN Du dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DSIM_LIT_FN V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DSIM_LIT_FN Du length(V3 a) {   // vec3.h:65-68, 194-197: zero gradient at |a| == 0
    const float l = sqrtf(a.x.v * a.x.v + a.y.v * a.y.v + a.z.v * a.z.v);
    if (l > 0.0f) return Du(l, (a.x.v * a.x.d + a.y.v * a.y.d + a.z.v * a.z.d) / l);
    return Du(l);
}
DSIM_LIT_FN V3 normalize(V3 a) {   // vec3.h:70-77
    const Du l = length(a);
    if (l.v > 0.0f) return V3(a.x / l, a.y / l, a.z / l);
    return V3();
}
struct Q4 {
    Du x, y, z, w;
    DSIM_LIT_FN Q4() : w(1.f) {}
    DSIM_LIT_FN Q4(Du a, Du b, Du c, Du d) : x(a), y(b), z(c), w(d) {}
};
DSIM_LIT_FN Q4 qmul(Q4 a, Q4 b) {   // quat.h:100-106
    return Q4(a.w * b.x + b.w * a.x + a.y * b.z - b.y * a.z, a.w * b.y + b.w * a.y + a.z * b.x - b.z * a.x,
              a.w * b.z + b.w * a.z + a.x * b.y - b.x * a.y, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
DSIM_LIT_FN V3 rotate(Q4 q, V3 x) {   // quat.h:113-116, evaluated literally (|q| may differ from 1 along the tangent)
    const V3 qv(q.x, q.y, q.z);
    return x * (Du(2.0f) * q.w * q.w - Du(1.0f)) + cross(qv, x) * q.w * Du(2.0f) + qv * dot(qv, x) * Du(2.0f);
}
DSIM_LIT_FN Q4 qnormalize(Q4 q) {   // quat.h:70-83
    const Du l = dsqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    if (l.v > 0.0f) {
        const Du il = Du(1.0f) / l;
        return Q4(q.x * il, q.y * il, q.z * il, q.w * il);
    }
    return Q4();
}
struct Xf {
    V3 p;
    Q4 q;
};
DSIM_LIT_FN Xf xf_mul(Xf a, Xf b) { return Xf{rotate(a.q, b.p) + a.p, qmul(a.q, b.q)}; }   // spatial.h:190-193
DSIM_LIT_FN V3 xf_point(Xf t, V3 x) { return t.p + rotate(t.q, x); }                        // spatial.h:209-212
struct SV {
    Du e[6];   // (w, v)
    DSIM_LIT_FN V3 w() const { return V3(e[0], e[1], e[2]); }
    DSIM_LIT_FN V3 v() const { return V3(e[3], e[4], e[5]); }
};
DSIM_LIT_FN SV mksv(V3 w, V3 v) {
    SV s;
    s.e[0] = w.x; s.e[1] = w.y; s.e[2] = w.z; s.e[3] = v.x; s.e[4] = v.y; s.e[5] = v.z;
    return s;
}
DSIM_LIT_FN SV operator+(SV a, SV b) {
    SV s;
    for (int k = 0; k < 6; ++k) s.e[k] = a.e[k] + b.e[k];
    return s;
}
DSIM_LIT_FN SV operator-(SV a, SV b) {
    SV s;
    for (int k = 0; k < 6; ++k) s.e[k] = a.e[k] - b.e[k];
    return s;
}
DSIM_LIT_FN SV operator*(SV a, Du x) {
    SV s;
    for (int k = 0; k < 6; ++k) s.e[k] = a.e[k] * x;
    return s;
}
DSIM_LIT_FN Du sdot(SV a, SV b) {
    Du s;
    for (int k = 0; k < 6; ++k) s = s + a.e[k] * b.e[k];
    return s;
}
DSIM_LIT_FN SV spatial_cross(SV a, SV b) { return mksv(cross(a.w(), b.w()), cross(a.v(), b.w()) + cross(a.w(), b.v())); }        // spatial.h:56-62
DSIM_LIT_FN SV spatial_cross_dual(SV a, SV b) { return mksv(cross(a.w(), b.w()) + cross(a.v(), b.v()), cross(a.w(), b.v())); }   // spatial.h:64-70
DSIM_LIT_FN SV transform_twist(Xf t, V3 w0, V3 v0) {   // sim.py:1076-1088
    const V3 w = rotate(t.q, w0);
    return mksv(w, rotate(t.q, v0) + cross(t.p, w));
}
struct M6 {
    Du d[6][6];
};
DSIM_LIT_FN SV m6vec(const M6& a, const SV& b) {   // spatial.h:548-557
    SV o;
    for (int i = 0; i < 6; ++i) {
        Du s;
        for (int j = 0; j < 6; ++j) s = s + a.d[i][j] * b.e[j];
        o.e[i] = s;
    }
    return o;
}
// sim.py:1105-1134 (spatial_transform_inverse + spatial_transform_inertia), I_m = diag(Ic, m 1) (dsim_model_create admits no other)
DSIM_LIT_FN void transform_inertia(Xf t, const float* ic6, float mass, M6& out) {
    const Q4 qi(-t.q.x, -t.q.y, -t.q.z, t.q.w);
    const V3 p = rotate(qi, t.p) * Du(-1.0f);
    const V3 r1 = rotate(qi, V3(Du(1.f), Du(0.f), Du(0.f))), r2 = rotate(qi, V3(Du(0.f), Du(1.f), Du(0.f))),
             r3 = rotate(qi, V3(Du(0.f), Du(0.f), Du(1.f)));
    const Du R[3][3] = {{r1.x, r2.x, r3.x}, {r1.y, r2.y, r3.y}, {r1.z, r2.z, r3.z}};   // mat33's column constructor
    const Du K[3][3] = {{Du(0.f), -p.z, p.y}, {p.z, Du(0.f), -p.x}, {-p.y, p.x, Du(0.f)}};
    Du S[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Du s;
            for (int k = 0; k < 3; ++k) s = s + K[i][k] * R[k][j];
            S[i][j] = s;
        }
    // A = [[R, 0], [S, R]] (spatial_adjoint, spatial.h:595-620); out = A^T I_m A
    const float Ic[3][3] = {{ic6[0], ic6[1], ic6[2]}, {ic6[1], ic6[3], ic6[4]}, {ic6[2], ic6[4], ic6[5]}};
    Du IR[3][3], mS[3][3], mR[3][3];   // Ic R, m S, m R
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Du s;
            for (int k = 0; k < 3; ++k) s = s + Du(Ic[i][k]) * R[k][j];
            IR[i][j] = s;
            mS[i][j] = Du(mass) * S[i][j];
            mR[i][j] = Du(mass) * R[i][j];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Du a, b, cc, dd;
            for (int k = 0; k < 3; ++k) {
                a = a + R[k][i] * IR[k][j] + S[k][i] * mS[k][j];   // upper left: R^T Ic R + S^T m S
                b = b + S[k][i] * mR[k][j];                         // upper right: S^T m R
                cc = cc + R[k][i] * mS[k][j];                       // lower left: R^T m S
                dd = dd + R[k][i] * mR[k][j];                       // lower right: R^T m R
            }
            out.d[i][j] = a;
            out.d[i][j + 3] = b;
            out.d[i + 3][j] = cc;
            out.d[i + 3][j + 3] = dd;
        }
}

// the model constants of one articulation as the layout builder packs them (dsim_layout.hpp: DsimLayout::cblob)
struct Consts {
    const uint32_t* cb;
    DsimOff o;
    DsimDims d;
    DSIM_LIT_FN int I(int off, int i) const { return (int)cb[off + i]; }
    DSIM_LIT_FN float F(int off, int i) const {
        union { uint32_t u; float f; } c;
        c.u = cb[off + i];
        return c.f;
    }
};

// rho of quaternion joint `jl` (a link index whose joint is FREE or BALL).  q, qd, act, mact: the state entering the first
// substep; h: substep length; hinv: the first group's H^-1 [nd][nd]; gq1, gqd1, aH: see the header comment.
DSIM_LIT_FN float radial(const Consts& c, const float* q, const float* qd, const float* act, const float* mact, float h,
                         const float* hinv, const float* gq1, const float* gqd1, const float* aH, int jl, float* dbg = nullptr) {
    const int L = c.d.L, nd = c.d.nd;
    const DsimOff& o = c.o;
    const int jtype = c.I(o.jtype, jl), jcs = c.I(o.qstart, jl) + (jtype == DSIM_JOINT_FREE ? 3 : 0);
    auto Q = [&](int k) { return (k >= jcs && k < jcs + 4) ? Du(q[k], q[k]) : Du(q[k]); };   // d/d eps of (1 + eps) q
    Xf X_sc[DSIM_LIT_LMAX];
    SV S_s[DSIM_LIT_NDMAX], v_s[DSIM_LIT_LMAX], a_s[DSIM_LIT_LMAX], f_s[DSIM_LIT_LMAX];
    M6 I_s[DSIM_LIT_LMAX];
    const V3 zero3;
    // ---- eval_rigid_fk + eval_rigid_id, link by link (parents precede children)
    const V3 g(Du(c.F(o.grav, 0)), Du(c.F(o.grav, 1)), Du(c.F(o.grav, 2)));
    for (int i = 0; i < L; ++i) {
        const int type = c.I(o.jtype, i), parent = c.I(o.parent, i), cs = c.I(o.qstart, i), ds = c.I(o.qdstart, i);
        const V3 axis(Du(c.F(o.axis, 3 * i)), Du(c.F(o.axis, 3 * i + 1)), Du(c.F(o.axis, 3 * i + 2)));
        Xf X_sp;
        if (parent >= 0) X_sp = X_sc[parent];
        Xf X_jc;
        if (type == DSIM_JOINT_PRISMATIC) {
            X_jc.p = axis * Q(cs);
        } else if (type == DSIM_JOINT_REVOLUTE) {   // quat_from_axis_angle, quat.h:44-52
            const Du half = Q(cs) * Du(0.5f);
            const Du cw(cosf(half.v), -sinf(half.v) * half.d), sn(sinf(half.v), cosf(half.v) * half.d);
            const V3 v = axis * sn;
            X_jc.q = Q4(v.x, v.y, v.z, cw);
        } else if (type == DSIM_JOINT_BALL) {
            X_jc.q = Q4(Q(cs), Q(cs + 1), Q(cs + 2), Q(cs + 3));
        } else if (type == DSIM_JOINT_FREE) {
            X_jc.p = V3(Q(cs), Q(cs + 1), Q(cs + 2));
            X_jc.q = Q4(Q(cs + 3), Q(cs + 4), Q(cs + 5), Q(cs + 6));
        }
        Xf X_pj;
        X_pj.p = V3(Du(c.F(o.xpj, 7 * i)), Du(c.F(o.xpj, 7 * i + 1)), Du(c.F(o.xpj, 7 * i + 2)));
        X_pj.q = Q4(Du(c.F(o.xpj, 7 * i + 3)), Du(c.F(o.xpj, 7 * i + 4)), Du(c.F(o.xpj, 7 * i + 5)), Du(c.F(o.xpj, 7 * i + 6)));
        X_sc[i] = xf_mul(X_sp, xf_mul(X_pj, X_jc));
        Xf X_cm;
        X_cm.p = V3(Du(c.F(o.com, 3 * i)), Du(c.F(o.com, 3 * i + 1)), Du(c.F(o.com, 3 * i + 2)));
        const Xf X_sm = xf_mul(X_sc[i], X_cm);
        const Xf X_sj = xf_mul(X_sp, X_pj);
        SV v_j;
        if (type == DSIM_JOINT_PRISMATIC) {
            S_s[ds] = transform_twist(X_sj, zero3, axis);
            v_j = S_s[ds] * Du(qd[ds]);
        } else if (type == DSIM_JOINT_REVOLUTE) {
            S_s[ds] = transform_twist(X_sj, axis, zero3);
            v_j = S_s[ds] * Du(qd[ds]);
        } else if (type == DSIM_JOINT_BALL) {
            S_s[ds] = transform_twist(X_sj, V3(Du(1.f), Du(0.f), Du(0.f)), zero3);
            S_s[ds + 1] = transform_twist(X_sj, V3(Du(0.f), Du(1.f), Du(0.f)), zero3);
            S_s[ds + 2] = transform_twist(X_sj, V3(Du(0.f), Du(0.f), Du(1.f)), zero3);
            v_j = S_s[ds] * Du(qd[ds]) + S_s[ds + 1] * Du(qd[ds + 1]) + S_s[ds + 2] * Du(qd[ds + 2]);
        } else if (type == DSIM_JOINT_FREE) {
            for (int k = 0; k < 6; ++k) {
                SV e;
                e.e[k] = Du(1.f);
                S_s[ds + k] = e;
                v_j.e[k] = Du(qd[ds + k]);
            }
        }
        SV v_p, a_p;
        if (parent >= 0) {
            v_p = v_s[parent];
            a_p = a_s[parent];
        }
        const SV v = v_p + v_j;
        const SV a = a_p + spatial_cross(v, v_j);
        const float mass = c.F(o.mass, i);
        // gravity wrench about the world origin: spatial_transform_wrench((X_sm.p, identity), (0, m g)), sim.py:1091-1103, 1772-1775
        const V3 fg = g * Du(mass);
        const SV f_g = mksv(cross(X_sm.p, fg), fg);
        float ic6[6];
        for (int k = 0; k < 6; ++k) ic6[k] = c.F(o.ic6, 6 * i + k);
        transform_inertia(X_sm, ic6, mass, I_s[i]);
        const SV f_b = m6vec(I_s[i], a) + spatial_cross_dual(v, m6vec(I_s[i], v));
        v_s[i] = v;
        a_s[i] = a;
        f_s[i] = f_b - f_g;
    }
    // ---- eval_rigid_contacts_art, sim.py:1137-1206
    for (int k = 0; k < c.d.C; ++k) {
        const int b = c.I(o.cbody, k);
        const V3 cp(Du(c.F(o.cpoint, 3 * k)), Du(c.F(o.cpoint, 3 * k + 1)), Du(c.F(o.cpoint, 3 * k + 2)));
        const Du cd(c.F(o.cdist, k)), ke(c.F(o.cmat, 4 * k)), kd(c.F(o.cmat, 4 * k + 1)), kf(c.F(o.cmat, 4 * k + 2)), mu(c.F(o.cmat, 4 * k + 3));
        const V3 n(Du(0.f), Du(1.f), Du(0.f));
        const V3 p = xf_point(X_sc[b], cp) - n * cd;
        const V3 dpdt = v_s[b].v() + cross(v_s[b].w(), p);
        const Du cc = dot(n, p);
        if (cc.v >= 0.0f) continue;
        const Du vn = dot(n, dpdt);
        const V3 vt = dpdt - n * vn;
        const Du fn = cc * ke;
        const Du fd = dmin(vn, Du(0.0f)) * kd * Du(dstep(cc)) * (Du(0.0f) - cc);
        const V3 ft = normalize(vt) * dmin(kf * length(vt), Du(0.0f) - mu * cc * ke) * Du(dstep(cc));
        const V3 ftot = n * (fn + fd) + ft;
        f_s[b] = f_s[b] + mksv(cross(p, ftot), ftot);
    }
    // ---- eval_muscles, sim.py:1209-1265 (the active segments: consecutive waypoints on different links)
    for (int s = 0; s < c.d.NS; ++s) {
        const int l0 = c.I(o.seg_rec, 8 * s) / 7, l1 = c.I(o.seg_rec, 8 * s + 1) / 7, wp = c.I(o.seg_rec, 8 * s + 2), mi = c.I(o.seg_rec, 8 * s + 3);
        const V3 r0(Du(c.F(o.mpoints, wp)), Du(c.F(o.mpoints, wp + 1)), Du(c.F(o.mpoints, wp + 2)));
        const V3 r1(Du(c.F(o.mpoints, wp + 3)), Du(c.F(o.mpoints, wp + 4)), Du(c.F(o.mpoints, wp + 5)));
        const V3 pos0 = xf_point(X_sc[l0], r0), pos1 = xf_point(X_sc[l1], r1);
        const V3 f = normalize(pos1 - pos0) * Du(mact[mi]);
        f_s[l0] = f_s[l0] - mksv(cross(pos0, f), f);
        f_s[l1] = f_s[l1] + mksv(cross(pos1, f), f);
    }
    // ---- eval_rigid_tau, reverse link order (f_s becomes the subtree total on the way), sim.py:1421-1502, 1792-1842
    Du tau[DSIM_LIT_NDMAX];
    for (int i = L - 1; i >= 0; --i) {
        const int type = c.I(o.jtype, i), parent = c.I(o.parent, i), cs = c.I(o.qstart, i), ds = c.I(o.qdstart, i);
        const Du tke(c.F(o.tke, i)), tkd(c.F(o.tkd, i)), lke(c.F(o.lke, i)), lkd(c.F(o.lkd, i));
        const SV f = f_s[i];
        if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
            const Du qq = Q(cs), qdv(qd[ds]), target(c.F(o.target, cs)), lower(c.F(o.lower, cs)), upper(c.F(o.upper, cs));
            Du limit_f(0.0f);
            if (qq.v < lower.v) limit_f = lke * (lower - qq);
            if (qq.v > upper.v) limit_f = lke * (upper - qq);
            tau[ds] = Du(0.0f) - sdot(S_s[ds], f) - tke * (qq - target) - tkd * qdv + Du(act[ds]) + limit_f + (Du(0.0f) - lkd) * qdv;
        } else if (type == DSIM_JOINT_BALL) {
            for (int k = 0; k < 3; ++k) tau[ds + k] = Du(0.0f) - sdot(S_s[ds + k], f) - Du(qd[ds + k]) * tkd - Q(cs + k) * tke;
        } else if (type == DSIM_JOINT_FREE) {
            for (int k = 0; k < 6; ++k) tau[ds + k] = Du(0.0f) - sdot(S_s[ds + k], f);
        }
        if (parent >= 0) f_s[parent] = f_s[parent] + f;
    }
    float rho = 0.f;
    // ---- < adj_H, d H >, H = J^T M J: H[a][b] = sum over the links i whose ancestors-or-self hold both dofs of S_a . I_s[i] S_b
    // (spatial.h:691-815 + the two gemms of sim.py:2514-2545 written out link by link; the armature has no tangent)
    for (int i = 0; i < L; ++i) {
        for (int ja = i; ja >= 0; ja = c.I(o.parent, ja)) {
            for (int a = c.I(o.qdstart, ja); a < c.I(o.qdstart, ja + 1); ++a) {
                const SV Ia = m6vec(I_s[i], S_s[a]);   // (I_s is symmetric: S_a . I S_b = (I S_a) . S_b)
                for (int jb = i; jb >= 0; jb = c.I(o.parent, jb))
                    for (int b = c.I(o.qdstart, jb); b < c.I(o.qdstart, jb + 1); ++b) {
                        const float dh = sdot(Ia, S_s[b]).d;
                        rho += aH[a * nd + b] * dh;
                        if (dbg) dbg[2 * nd + a * nd + b] += dh;   // (host tests: tangents of tau, qdd, H)
                    }
            }
        }
    }
    // ---- d qdd = H^-1 d tau (H held fixed: its own tangent went through adj_H above)
    float dqdd[DSIM_LIT_NDMAX];
    for (int i = 0; i < nd; ++i) {
        float s = 0.f;
        for (int j = 0; j < nd; ++j) s += hinv[i * nd + j] * tau[j].d;
        dqdd[i] = s;
        if (dbg) {
            dbg[i] = tau[i].d;
            dbg[nd + i] = s;
        }
    }
    // ---- eval_rigid_integrate (sim.py:1505-1636) on (value: the checkpointed H^-1 tau, tangent: d qdd), contracted with gq_1 / gqd_1
    const Du hh(h);
    for (int i = 0; i < L; ++i) {
        const int type = c.I(o.jtype, i), cs = c.I(o.qstart, i), ds = c.I(o.qdstart, i);
        auto QDD = [&](int k) {
            float s = 0.f;
            for (int j = 0; j < nd; ++j) s += hinv[k * nd + j] * tau[j].v;
            return Du(s, dqdd[k]);
        };
        if (type == DSIM_JOINT_PRISMATIC || type == DSIM_JOINT_REVOLUTE) {
            const Du qdn = Du(qd[ds]) + QDD(ds) * hh;
            const Du qn = Q(cs) + qdn * hh;
            rho += gqd1[ds] * qdn.d + gq1[cs] * qn.d;
        } else if (type == DSIM_JOINT_BALL || type == DSIM_JOINT_FREE) {
            const bool fr = type == DSIM_JOINT_FREE;
            const V3 w = V3(Du(qd[ds]), Du(qd[ds + 1]), Du(qd[ds + 2])) + V3(QDD(ds), QDD(ds + 1), QDD(ds + 2)) * hh;
            const int rs = cs + (fr ? 3 : 0);
            if (fr) {
                const V3 v = V3(Du(qd[ds + 3]), Du(qd[ds + 4]), Du(qd[ds + 5])) + V3(QDD(ds + 3), QDD(ds + 4), QDD(ds + 5)) * hh;
                const V3 p(Q(cs), Q(cs + 1), Q(cs + 2));
                const V3 pn = p + (v + cross(w, p)) * hh;
                rho += gq1[cs] * pn.x.d + gq1[cs + 1] * pn.y.d + gq1[cs + 2] * pn.z.d;
                rho += gqd1[ds + 3] * v.x.d + gqd1[ds + 4] * v.y.d + gqd1[ds + 5] * v.z.d;
            }
            const Q4 r(Q(rs), Q(rs + 1), Q(rs + 2), Q(rs + 3));
            Q4 dr = qmul(Q4(w.x, w.y, w.z, Du(0.0f)), r);
            dr = Q4(dr.x * Du(0.5f), dr.y * Du(0.5f), dr.z * Du(0.5f), dr.w * Du(0.5f));
            const Q4 rn = qnormalize(Q4(r.x + dr.x * hh, r.y + dr.y * hh, r.z + dr.z * hh, r.w + dr.w * hh));
            rho += gq1[rs] * rn.x.d + gq1[rs + 1] * rn.y.d + gq1[rs + 2] * rn.z.d + gq1[rs + 3] * rn.w.d;
            rho += gqd1[ds] * w.x.d + gqd1[ds + 1] * w.y.d + gqd1[ds + 2] * w.z.d;
        }
    }
    return rho;
}

}  // namespace dsim_lit
