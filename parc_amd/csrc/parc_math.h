// Device quaternion / rotation math for the tracker kernels (gfx950).
// Quaternions are xyzw, fp32, as in the reference's util/torch_util.py; each helper follows the
// reference's operation order so results agree with it to a few ulp.
#pragma once
#include <hip/hip_runtime.h>

#define PARC_DEV __device__ __forceinline__

struct q4 {
    float x, y, z, w;
};
struct v3 {
    float x, y, z;
};

PARC_DEV v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
PARC_DEV q4 mk4(float x, float y, float z, float w) { return q4{x, y, z, w}; }
PARC_DEV v3 operator+(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PARC_DEV v3 operator-(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PARC_DEV v3 operator*(float s, v3 a) { return v3{s * a.x, s * a.y, s * a.z}; }
PARC_DEV float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PARC_DEV v3 cross3(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// ---- single-instruction / short-polynomial primitives (each within ~1-2 ulp of the correctly rounded value; the
// library versions cost 10-200 instructions per call and the post-step kernel is VALU-issue bound) -----------------
PARC_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
PARC_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PARC_DEV float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// sin and cos of x, |x| up to a few hundred: k = rint(x * 2/pi), two-constant reduction by pi/2, degree-7/8
// polynomials on [-pi/4, pi/4] (least-squares fits, |err| < 8e-8), quadrant fix-up
PARC_DEV void fsincos(float x, float &s, float &c) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.5707963705062866f, x);
    r = fmaf(k, 4.3711388286737929e-08f, r);
    float z = r * r;
    float ps = fmaf(z, fmaf(z, -0.0001947956479853019f, 0.0083318455144763f), -0.16666647791862488f);
    float pc = fmaf(z, fmaf(z, 2.4421184207312763e-05f, -0.001388721400871873f), 0.04166664183139801f);
    float sr = fmaf(r * z, ps, r);
    float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    int q = (int)k;
    float so = (q & 1) ? cr : sr;
    float co = (q & 1) ? sr : cr;
    s = (q & 2) ? -so : so;
    c = ((q + 1) & 2) ? -co : co;
}

// atan2(y, x) for y >= 0, x >= 0 (the only case the quaternion angle needs): odd degree-17 fit on [0, 1], |err| < 8e-8
PARC_DEV float fatan2_q1(float y, float x) {
    float mn = fminf(x, y), mx = fmaxf(x, y);
    float t = mn * frcp(mx);
    t = mx > 0.f ? t : 0.f;
    float z = t * t;
    float p = fmaf(z, -0.0025300427805632353f, 0.014093323610723019f);
    p = fmaf(z, p, -0.036850083619356155f);
    p = fmaf(z, p, 0.06335900723934174f);
    p = fmaf(z, p, -0.08698903024196625f);
    p = fmaf(z, p, 0.11045123636722565f);
    p = fmaf(z, p, -0.1428011804819107f);
    p = fmaf(z, p, 0.19999824464321136f);
    p = fmaf(z, p, -0.3333333134651184f);
    float a = fmaf(t * z, p, t);
    return y > x ? 1.5707963267948966f - a : a;
}

// acos(c) for 0 <= c <= 1: sqrt(1 - c) * P(c), degree-7 fit, relative error < 2.1e-7
PARC_DEV float facos01(float c) {
    float p = fmaf(c, -0.0012370048789307475f, 0.006580885034054518f);
    p = fmaf(c, p, -0.01696547120809555f);
    p = fmaf(c, p, 0.030808253213763237f);
    p = fmaf(c, p, -0.05014502629637718f);
    p = fmaf(c, p, 0.08897409588098526f);
    p = fmaf(c, p, -0.21459849178791046f);
    p = fmaf(c, p, 1.570796251296997f);
    return fsqrt(fmaxf(1.0f - c, 0.f)) * p;
}

// util/torch_util.py:4-7: atan2(sin x, cos x) = x wrapped to (-pi, pi]
PARC_DEV float normalize_angle(float x) { return fmaf(rintf(x * 0.15915494309189535f), -6.283185307179586f, x); }

// util/torch_util.py:40-58.  The reference evaluates the product with the 9-multiplication / 27-addition grouping; this is
// the same bilinear form in the 16-FMA grouping (identical in exact arithmetic for any a, b - unit or not -, a few ulp
// apart in fp32, and 13 instructions shorter; the forward-kinematics sweep issues it once per tree level).
PARC_DEV q4 quat_mul(q4 a, q4 b) {
    q4 o;
    o.x = fmaf(a.w, b.x, fmaf(a.x, b.w, fmaf(a.y, b.z, -a.z * b.y)));
    o.y = fmaf(a.w, b.y, fmaf(a.y, b.w, fmaf(a.z, b.x, -a.x * b.z)));
    o.z = fmaf(a.w, b.z, fmaf(a.z, b.w, fmaf(a.x, b.y, -a.y * b.x)));
    o.w = fmaf(a.w, b.w, -fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)));
    return o;
}

// util/torch_util.py:60-66
PARC_DEV v3 quat_rotate(q4 q, v3 v) {
    v3 qv = mk3(q.x, q.y, q.z);
    v3 t = 2.f * cross3(qv, v);
    v3 c = cross3(qv, t);
    return v3{v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}

PARC_DEV q4 quat_conj(q4 q) { return q4{-q.x, -q.y, -q.z, q.w}; }

// util/torch_util.py:33-38
PARC_DEV q4 quat_pos(q4 q) {
    float s = q.w < 0.f ? -1.f : 1.f;
    return q4{s * q.x, s * q.y, s * q.z, s * q.w};
}

// util/torch_util.py:9-12 (eps 1e-9) on a quaternion
PARC_DEV q4 quat_unit(q4 q) {
    float n = fsqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float inv = frcp(fmaxf(n, 1e-9f));
    return q4{q.x * inv, q.y * inv, q.z * inv, q.w * inv};
}

// util/torch_util.py:311-317
PARC_DEV q4 axis_angle_to_quat(v3 axis, float angle) {
    float theta = angle * 0.5f;
    float s, c;
    fsincos(theta, s, c);
    s *= frcp(fmaxf(fsqrt(dot3(axis, axis)), 1e-9f));
    return quat_unit(q4{axis.x * s, axis.y * s, axis.z * s, c});
}

// sin on [0, pi/2] without range reduction: odd Taylor polynomial through x^13 (truncation < 6e-8 at pi/2);
// the slerp weights only ever need this interval (half angle in [0, pi/2], blend in [0, 1])
PARC_DEV float sin_0_halfpi(float x) {
    float x2 = x * x;
    float p = fmaf(x2, 1.6059043836821613e-10f, -2.5052108385441720e-08f);
    p = fmaf(x2, p, 2.7557319223985893e-06f);
    p = fmaf(x2, p, -1.9841269841269841e-04f);
    p = fmaf(x2, p, 8.3333333333333333e-03f);
    p = fmaf(x2, p, -1.6666666666666666e-01f);
    return fmaf(x * x2, p, x);
}

// util/torch_util.py:394-419
PARC_DEV q4 exp_map_to_quat(v3 em) {
    float a = fsqrt(dot3(em, em));
    float ia = frcp(a);
    v3 ax = v3{em.x * ia, em.y * ia, em.z * ia};
    if (!(a < 3.1415925f)) a = normalize_angle(a);       // identity below pi
    bool ok = fabsf(a) > 1e-5f;
    if (!ok) {
        ax = mk3(0.f, 0.f, 1.f);
        a = 0.f;
    }
    return axis_angle_to_quat(ax, a);
}

// Library-precision variants for the one-time clip database build: the stored frame quaternions then equal the
// reference's to the last bit, which matters because slerp between nearly identical frames amplifies input ulps.
PARC_DEV q4 quat_unit_lib(q4 q) {
#pragma clang fp contract(off)      // op by op like torch: no fused multiply-adds in the *_lib functions
    float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    return q4{q.x / n, q.y / n, q.z / n, q.w / n};
}
PARC_DEV q4 axis_angle_to_quat_lib(v3 axis, float angle) {
#pragma clang fp contract(off)
    float theta = angle / 2.f;
    float n = fmaxf(sqrtf((axis.x * axis.x + axis.y * axis.y) + axis.z * axis.z), 1e-9f);
    float s = sinf(theta);
    return quat_unit_lib(q4{axis.x / n * s, axis.y / n * s, axis.z / n * s, cosf(theta)});
}
PARC_DEV q4 exp_map_to_quat_lib(v3 em) {
#pragma clang fp contract(off)
    float a = sqrtf((em.x * em.x + em.y * em.y) + em.z * em.z);
    v3 ax = v3{em.x / a, em.y / a, em.z / a};
    a = atan2f(sinf(a), cosf(a));
    if (!(fabsf(a) > 1e-5f)) {
        ax = mk3(0.f, 0.f, 1.f);
        a = 0.f;
    }
    return axis_angle_to_quat_lib(ax, a);
}

// util/torch_util.py:68-88
PARC_DEV void quat_to_axis_angle(q4 qin, v3 &axis, float &angle) {
    q4 q = quat_pos(qin);
    float len = fsqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    float a = 2.0f * fatan2_q1(len, q.w);                // q.w >= 0 after quat_pos
    if (len > 1e-5f) {
        float il = frcp(len);
        axis = v3{q.x * il, q.y * il, q.z * il};
        angle = a;
    } else {
        axis = mk3(0.f, 0.f, 1.f);
        angle = 0.f;
    }
}

// util/torch_util.py:346-351
PARC_DEV v3 quat_to_exp_map(q4 q) {
    v3 ax;
    float an;
    quat_to_axis_angle(q, ax, an);
    return an * ax;
}

// util/torch_util.py:427-431
PARC_DEV float quat_diff_angle(q4 q0, q4 q1) {
    v3 ax;
    float an;
    quat_to_axis_angle(quat_mul(q1, quat_conj(q0)), ax, an);
    return an;
}

// sum of the four rounded products, left to right (no fma contraction): slerp between nearly identical frames amplifies one
// ulp of this cosine by 1/sin^2, so it has to be evaluated the way the reference (and the oracle) evaluates it
PARC_DEV float dot4_unfused(q4 a, q4 b) {
#pragma clang fp contract(off)
    float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z, p3 = a.w * b.w;
    return ((p0 + p1) + p2) + p3;
}

// 1 - c*c with the product rounded first (no fma contraction), as torch evaluates it
PARC_DEV float one_minus_sq_unfused(float c) {
#pragma clang fp contract(off)
    float cc = c * c;
    return 1.0f - cc;
}

// a * b + c with the product rounded before the sum (two roundings, the way torch evaluates `x * k + c` on tensors)
PARC_DEV float mul_add_unfused(float a, float b, float c) {
#pragma clang fp contract(off)
    float p = a * b;
    return p + c;
}

// util/torch_util.py:443-468
PARC_DEV q4 slerp(q4 q0, q4 q1, float t) {
    float c = dot4_unfused(q0, q1);
    float sg = c < 0.f ? -1.f : 1.f;
    q1 = q4{sg * q1.x, sg * q1.y, sg * q1.z, sg * q1.w};
    c = fabsf(c);
    float ht = facos01(fminf(c, 1.0f));
    float s = fsqrt(one_minus_sq_unfused(c));   // the reference rounds c*c before the subtraction, and s amplifies it for close frames
    float is = frcp(s);
    float ra = sin_0_halfpi((1.f - t) * ht) * is;
    float rb = sin_0_halfpi(t * ht) * is;
    q4 o = q4{ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w};
    if (fabsf(s) < 0.001f) o = q4{0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z, 0.5f * q0.w + 0.5f * q1.w};
    if (fabsf(c) >= 1.f) o = q0;
    return o;
}

// util/torch_util.py:470-479
PARC_DEV float calc_heading(q4 q) {
    v3 d = quat_rotate(q, mk3(1.f, 0.f, 0.f));
    return atan2f(d.y, d.x);
}

// util/torch_util.py:491-499
PARC_DEV q4 calc_heading_quat_inv(q4 q) { return axis_angle_to_quat(mk3(0.f, 0.f, 1.f), -calc_heading(q)); }

// Same rotation without atan2/sin/cos: the heading h has cos h = a/r, sin h = b/r for the rotated x axis (a, b, .);
// the half-angle values follow from the numerically stable branch of the half-angle formulas.
PARC_DEV q4 calc_heading_quat_inv_alg(q4 q) {
    float a = 1.0f - 2.0f * (q.y * q.y + q.z * q.z);
    float b = 2.0f * (q.w * q.z + q.x * q.y);
    float r2 = a * a + b * b;
    if (!(r2 > 0.f)) return mk4(0.f, 0.f, 0.f, 1.f);
    float ir = __builtin_amdgcn_rsqf(r2);
    float ch = a * ir, sh = b * ir;
    float c2, s2;
    if (ch >= 0.f) {
        c2 = fsqrt(0.5f * (1.0f + ch));
        s2 = sh * frcp(2.0f * c2);
    } else {
        s2 = (sh >= 0.f ? 1.f : -1.f) * fsqrt(0.5f * (1.0f - ch));
        c2 = sh * frcp(2.0f * s2);
    }
    return mk4(0.f, 0.f, -s2, c2);   // rotation by -h about z
}

// util/torch_util.py:361-373: 6 floats = R(q) e_x | R(q) e_z.  quat_rotate with the unit vectors substituted (the
// products with the zero components dropped; same operation order for the surviving terms)
PARC_DEV void quat_to_tan_norm(q4 q, float *o) {
    float ty = 2.f * q.z, tz = -2.f * q.y;          // t = 2 qv x e_x = (0, 2 qz, -2 qy)
    o[0] = 1.f + (q.y * tz - q.z * ty);
    o[1] = q.w * ty - q.x * tz;
    o[2] = q.w * tz + q.x * ty;
    float ux = 2.f * q.y, uy = -2.f * q.x;          // u = 2 qv x e_z = (2 qy, -2 qx, 0)
    o[3] = q.w * ux - q.z * uy;
    o[4] = q.w * uy + q.z * ux;
    o[5] = 1.f + (q.x * uy - q.y * ux);
}

PARC_DEV q4 ld4(const float *p) { return q4{p[0], p[1], p[2], p[3]}; }
PARC_DEV v3 ld3(const float *p) { return v3{p[0], p[1], p[2]}; }
PARC_DEV void st4(float *p, q4 q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
PARC_DEV void st3(float *p, v3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
