// Developer experiment (round 6; no GPU needed: hipcc -S and count): the world-inertia + body-force block of the Ant forward kernel
// (dsim_core.hpp: dsim_fwd_kinematics_walk_mid behind the hand-over) in its scalar form and hand-packed on (xy pair, z) 3-vectors.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S pk_block_test.hip; result in profiles/r06_experiments.txt item 5.
#include <hip/hip_runtime.h>
#define DSIM_FN __device__ __forceinline__
#include "../../diffrl_amd/csrc/dsim_math.hpp"

typedef float f2 __attribute__((ext_vector_type(2)));
struct p3 { f2 xy; float z; };   // packed 3-vector
DSIM_FN p3 P3(v3 a) { return p3{f2{a.x, a.y}, a.z}; }
DSIM_FN p3 operator+(p3 a, p3 b) { return p3{a.xy + b.xy, a.z + b.z}; }
DSIM_FN p3 operator-(p3 a, p3 b) { return p3{a.xy - b.xy, a.z - b.z}; }
DSIM_FN p3 operator*(p3 a, float s) { return p3{a.xy * s, a.z * s}; }
DSIM_FN p3 fma3(p3 a, float s, p3 c) { return p3{__builtin_elementwise_fma(a.xy, f2{s, s}, c.xy), __builtin_fmaf(a.z, s, c.z)}; }
// cross with packed ops: (c.x, -c.y) = (a.y b.z - a.z b.y, a.x b.z - a.z b.x); c.z = a.x b.y - a.y b.x
DSIM_FN p3 pcross(p3 a, p3 b) {
    f2 ayx = __builtin_shufflevector(a.xy, a.xy, 1, 0), byx = __builtin_shufflevector(b.xy, b.xy, 1, 0);
    f2 t = ayx * f2{b.z, b.z} - f2{a.z, a.z} * byx;   // (a.y b.z - a.z b.y, a.x b.z - a.z b.x)
    return p3{f2{t.x, -t.y}, a.xy.x * b.xy.y - a.xy.y * b.xy.x};
}

extern "C" __global__ void k_scalar(const float* in, float* out) {
    const int t = threadIdx.x;
    const float* p = in + 64 * t;
    q4 rc = ldq(p); v3 com = ld3(p + 4), pc = ld3(p + 7), grav = ld3(p + 10);
    float ic0 = p[13], ic1 = p[14], ic2 = p[15], ic3 = p[16], ic4 = p[17], ic5 = p[18], m = p[19];
    sv6 wa = ldsv(p + 20), wv = ldsv(p + 26);
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
    const sv6 fb = inertia_mul(I, wa) + scross_dual(wv, inertia_mul(I, wv));
    const v3 mg = grav * m;
    const sv6 fg = mksv(cross(cm, mg), mg);
    float* o = out + 32 * t;
    st_i10(o, I);
    stsv(o + 10, fb - fg);
}

// the 6-vector part only: fb = I a + v x* (I v), packed
DSIM_FN void pinertia_mul(const inertia10& I, p3 xw, p3 xv, p3& yw, p3& yv) {
    // sym_mul as a combination of A's columns
    p3 c0{f2{I.axx, I.axy}, I.axz}, c1{f2{I.axy, I.ayy}, I.ayz}, c2{f2{I.axz, I.ayz}, I.azz};
    p3 h = P3(I.h);
    yw = fma3(c2, xw.z, fma3(c1, xw.xy.y, c0 * xw.xy.x)) + pcross(h, xv);
    yv = xv * I.m + pcross(xw, h);
}
extern "C" __global__ void k_packed(const float* in, float* out) {
    const int t = threadIdx.x;
    const float* p = in + 64 * t;
    q4 rc = ldq(p); v3 com = ld3(p + 4), pc = ld3(p + 7), grav = ld3(p + 10);
    float ic0 = p[13], ic1 = p[14], ic2 = p[15], ic3 = p[16], ic4 = p[17], ic5 = p[18], m = p[19];
    sv6 wa = ldsv(p + 20), wv = ldsv(p + 26);
    const v3 cm = rotate(rc, com) + pc;
    v3 rx, ry, rz;
    rotate_basis(rc, rx, ry, rz);
    p3 prx = P3(rx), pry = P3(ry), prz = P3(rz);
    const p3 b0 = fma3(prz, ic2, fma3(pry, ic1, prx * ic0));
    const p3 b1 = fma3(prz, ic4, fma3(pry, ic3, prx * ic1));
    const p3 b2 = fma3(prz, ic5, fma3(pry, ic4, prx * ic2));
    inertia10 I;
    I.m = m;
    I.h = cm * m;
    const float cc = dot(cm, cm);
    // rows of B R^T: row_i = b0[i] rx + b1[i] ry + b2[i] rz
    const p3 r0 = fma3(prz, b2.xy.x, fma3(pry, b1.xy.x, prx * b0.xy.x));
    const p3 r1 = fma3(prz, b2.xy.y, fma3(pry, b1.xy.y, prx * b0.xy.y));
    const float r2z = b0.z * rz.x * 0.f + b0.z * rx.z + b1.z * ry.z + b2.z * rz.z;
    I.axx = r0.xy.x + m * (cc - cm.x * cm.x);
    I.axy = r0.xy.y - m * cm.x * cm.y;
    I.axz = r0.z - m * cm.x * cm.z;
    I.ayy = r1.xy.y + m * (cc - cm.y * cm.y);
    I.ayz = r1.z - m * cm.y * cm.z;
    I.azz = r2z + m * (cc - cm.z * cm.z);
    p3 aw = P3(wa.w), av = P3(wa.v), vw = P3(wv.w), vv = P3(wv.v);
    p3 Iaw, Iav, Ivw, Ivv;
    pinertia_mul(I, aw, av, Iaw, Iav);
    pinertia_mul(I, vw, vv, Ivw, Ivv);
    // scross_dual(v, Iv) = (vw x Ivw + vv x Ivv, vw x Ivv)
    p3 fw = Iaw + pcross(vw, Ivw) + pcross(vv, Ivv), fv = Iav + pcross(vw, Ivv);
    const v3 mg = grav * m;
    const sv6 fg = mksv(cross(cm, mg), mg);
    float* o = out + 32 * t;
    st_i10(o, I);
    o[10] = fw.xy.x - fg.w.x; o[11] = fw.xy.y - fg.w.y; o[12] = fw.z - fg.w.z;
    o[13] = fv.xy.x - fg.v.x; o[14] = fv.xy.y - fg.v.y; o[15] = fv.z - fg.v.z;
}
