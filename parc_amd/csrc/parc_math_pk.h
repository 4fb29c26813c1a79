// Two poses per lane: the helpers of parc_math.h on float2 values, component i = pose i.
// gfx950 issues v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 at the rate of their scalar forms, so a lane that carries two poses pays
// one vector instruction for the multiply-adds of both (the post-step kernel is bound by vector issue).  Every
// function below performs, per component, exactly the operations of its scalar namesake in the same order (same polynomial, same
// unfused products where the scalar form asks for them), so a packed pose equals the scalar one up to the compiler's choice of which
// product of an `a*b + c*d` it fuses.  What has no packed instruction (v_rcp, v_sqrt, compares / selects, cross-lane moves) is issued
// per component.
#pragma once
#include "parc_math.h"

typedef float f2 __attribute__((ext_vector_type(2)));

struct q4p {
    f2 x, y, z, w;
};
struct v3p {
    f2 x, y, z;
};

PARC_DEV f2 sp2(float a) { return f2{a, a}; }
PARC_DEV f2 mk2(float a, float b) { return f2{a, b}; }
PARC_DEV q4p sp4(q4 q) { return q4p{sp2(q.x), sp2(q.y), sp2(q.z), sp2(q.w)}; }
PARC_DEV v3p sp3(v3 v) { return v3p{sp2(v.x), sp2(v.y), sp2(v.z)}; }
PARC_DEV q4p pair4(q4 a, q4 b) { return q4p{mk2(a.x, b.x), mk2(a.y, b.y), mk2(a.z, b.z), mk2(a.w, b.w)}; }
PARC_DEV v3p pair3(v3 a, v3 b) { return v3p{mk2(a.x, b.x), mk2(a.y, b.y), mk2(a.z, b.z)}; }
PARC_DEV f2 pfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
PARC_DEV f2 pmax(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
PARC_DEV f2 pmin(f2 a, f2 b) { return f2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
PARC_DEV f2 pabs(f2 a) { return f2{fabsf(a.x), fabsf(a.y)}; }
PARC_DEV f2 psqrt(f2 a) { return f2{fsqrt(a.x), fsqrt(a.y)}; }
PARC_DEV f2 prcp(f2 a) { return f2{frcp(a.x), frcp(a.y)}; }

PARC_DEV v3p operator+(v3p a, v3p b) { return v3p{a.x + b.x, a.y + b.y, a.z + b.z}; }
PARC_DEV v3p operator-(v3p a, v3p b) { return v3p{a.x - b.x, a.y - b.y, a.z - b.z}; }
PARC_DEV v3p operator*(float s, v3p a) { return v3p{s * a.x, s * a.y, s * a.z}; }
PARC_DEV v3p cross3(v3p a, v3p b) { return v3p{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

PARC_DEV q4p quat_mul(q4p a, q4p b) {
    q4p o;
    o.x = pfma(a.w, b.x, pfma(a.x, b.w, pfma(a.y, b.z, -a.z * b.y)));
    o.y = pfma(a.w, b.y, pfma(a.y, b.w, pfma(a.z, b.x, -a.x * b.z)));
    o.z = pfma(a.w, b.z, pfma(a.z, b.w, pfma(a.x, b.y, -a.y * b.x)));
    o.w = pfma(a.w, b.w, -pfma(a.x, b.x, pfma(a.y, b.y, a.z * b.z)));
    return o;
}

PARC_DEV v3p quat_rotate(q4p q, v3p v) {
    v3p qv = v3p{q.x, q.y, q.z};
    v3p t = 2.f * cross3(qv, v);
    v3p c = cross3(qv, t);
    return v3p{v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}

PARC_DEV f2 facos01(f2 c) {
    f2 p = pfma(c, sp2(-0.0012370048789307475f), sp2(0.006580885034054518f));
    p = pfma(c, p, sp2(-0.01696547120809555f));
    p = pfma(c, p, sp2(0.030808253213763237f));
    p = pfma(c, p, sp2(-0.05014502629637718f));
    p = pfma(c, p, sp2(0.08897409588098526f));
    p = pfma(c, p, sp2(-0.21459849178791046f));
    p = pfma(c, p, sp2(1.570796251296997f));
    return psqrt(pmax(1.0f - c, sp2(0.f))) * p;
}

PARC_DEV f2 sin_0_halfpi(f2 x) {
    f2 x2 = x * x;
    f2 p = pfma(x2, sp2(1.6059043836821613e-10f), sp2(-2.5052108385441720e-08f));
    p = pfma(x2, p, sp2(2.7557319223985893e-06f));
    p = pfma(x2, p, sp2(-1.9841269841269841e-04f));
    p = pfma(x2, p, sp2(8.3333333333333333e-03f));
    p = pfma(x2, p, sp2(-1.6666666666666666e-01f));
    return pfma(x * x2, p, x);
}

PARC_DEV f2 dot4_unfused(q4p a, q4p b) {
#pragma clang fp contract(off)
    f2 p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z, p3 = a.w * b.w;
    return ((p0 + p1) + p2) + p3;
}

PARC_DEV f2 one_minus_sq_unfused(f2 c) {
#pragma clang fp contract(off)
    f2 cc = c * c;
    return 1.0f - cc;
}

// util/torch_util.py:443-468, per component as slerp() of parc_math.h.  The two special cases replace the result by 0.5 q0 + 0.5 q1 and
// by q0: both are the general blend ra q0 + rb q1 with other weights (0.5, 0.5 and 1, 0 - exact), so they select the weights.
PARC_DEV q4p slerp(q4p q0, q4p q1, f2 t) {
    f2 c = dot4_unfused(q0, q1);
    const f2 sg = f2{c.x < 0.f ? -1.f : 1.f, c.y < 0.f ? -1.f : 1.f};
    q1 = q4p{sg * q1.x, sg * q1.y, sg * q1.z, sg * q1.w};
    c = pabs(c);
    f2 ht = facos01(pmin(c, sp2(1.0f)));
    f2 s = psqrt(one_minus_sq_unfused(c));
    f2 is = prcp(s);
    f2 ra = sin_0_halfpi((1.f - t) * ht) * is;
    f2 rb = sin_0_halfpi(t * ht) * is;
    const bool avg0 = fabsf(s.x) < 0.001f, avg1 = fabsf(s.y) < 0.001f, one0 = fabsf(c.x) >= 1.f, one1 = fabsf(c.y) >= 1.f;
    ra = f2{one0 ? 1.f : (avg0 ? 0.5f : ra.x), one1 ? 1.f : (avg1 ? 0.5f : ra.y)};
    rb = f2{one0 ? 0.f : (avg0 ? 0.5f : rb.x), one1 ? 0.f : (avg1 ? 0.5f : rb.y)};
    return q4p{ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w};
}

PARC_DEV f2 lerp_ref(f2 a, f2 b, f2 t) { return (1.0f - t) * a + t * b; }

// util/torch_util.py:361-373 (quat_to_tan_norm of parc_math.h): 6 values per pose
PARC_DEV void quat_to_tan_norm(q4p q, f2 *o) {
    f2 ty = 2.f * q.z, tz = -2.f * q.y;
    o[0] = 1.f + (q.y * tz - q.z * ty);
    o[1] = q.w * ty - q.x * tz;
    o[2] = q.w * tz + q.x * ty;
    f2 ux = 2.f * q.y, uy = -2.f * q.x;
    o[3] = q.w * ux - q.z * uy;
    o[4] = q.w * uy + q.z * ux;
    o[5] = 1.f + (q.x * uy - q.y * ux);
}
