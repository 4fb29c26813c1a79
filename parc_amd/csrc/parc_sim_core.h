// Articulated-body dynamics of the tracking humanoid: one environment per lane.
//
// What the reference delegates to Isaac Gym / PhysX (envs/ig_env.py:830-837 `gym.simulate`, articulation from
// data/assets/humanoid.xml, PD position drives envs/ig_char_env.py:115-135,489-495, terrain trimesh
// util/ig_util.py:6-22) is re-designed here from scratch -- there is no arithmetic to match (parity unpinned,
// DESIGN.md):
//   * reduced coordinates, Featherstone's articulated-body algorithm (O(n), exact joints), floating base;
//     spherical joints carry a quaternion, their dofs are exponential maps / child-frame angular velocity,
//     the convention of the reference's kinematics (anim/kin_char_model.py:57-100,552-581);
//   * implicit ("stable") PD: the drive's stiffness and damping enter the joint-space inertia
//     D_i += armature + h*kd + h^2*kp, so the 1000 N*m/rad / 100 N*m*s/rad gains are stable at h = 1/120 s;
//     torque limits (motor gears) scale the drive down when it saturates; joint limits are one-sided
//     implicit spring-dampers of the same form;
//   * contact against the heightfield taken as flat-topped columns with vertical walls (what the reference's
//     voxel mesh encodes, util/terrain_util.py:1099-1251): collision geometry = sample spheres on every geom
//     (box corners have radius 0); each penetrating sample contributes an implicit spring-damper + regularised
//     Coulomb friction, folded into the body's articulated inertia (dI = h * J^T Z J) -- unconditionally
//     stable, O(n), no iteration;
//   * semi-implicit Euler, `substeps` per call.
// The same source compiles for the device (hipcc, one thread = one env, per-thread arrays live in scratch,
// which the hardware interleaves per lane: every access is coalesced over the env axis) and for the host
// (g++: invariants / sanitizer build used by tests only).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/parc_sim.h"

#if defined(__HIPCC__)
#define PARC_HD __host__ __device__ __forceinline__
#else
#define PARC_HD static inline
#endif

// PARC_LOOP(k): no-op in the product.  tests/tools/bisect_sim_o3.py rebuilds the simulator sources with -DPARC_BISECT -DPARC_ROLL_<k> (and
// -I tests/tools, where the macro table parc_sim_bisect.h lives) to keep the loops tagged k rolled at -O3, which is how the -O3 divergence of
// sim_step_kernel was localised.
#if defined(PARC_BISECT)
#include "parc_sim_bisect.h"
#else
#define PARC_LOOP(k)
#endif

namespace parc_sim {

// Scalar primitives.  Device: the hardware's 1-ulp sqrt / reciprocal and short polynomials (the kernel is one long
// dependent instruction stream per wave; the library versions are 10-200 instructions each).  Host build: libm.
#if defined(__HIP_DEVICE_COMPILE__)
PARC_HD float p_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PARC_HD float p_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
PARC_HD void p_sincos(float x, float &s, float &c) {
    float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.5707963705062866f, x);
    r = fmaf(k, 4.3711388286737929e-08f, r);
    float z = r * r;
    float ps = fmaf(z, fmaf(z, -0.0001947956479853019f, 0.0083318455144763f), -0.16666647791862488f);
    float pc = fmaf(z, fmaf(z, 2.4421184207312763e-05f, -0.001388721400871873f), 0.04166664183139801f);
    float sr = fmaf(r * z, ps, r);
    float cr = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    int q = (int)k;
    float so = (q & 1) ? cr : sr, co = (q & 1) ? sr : cr;
    s = (q & 2) ? -so : so;
    c = ((q + 1) & 2) ? -co : co;
}
// atan2(y, x) for y >= 0, x >= 0
PARC_HD float p_atan2_q1(float y, float x) {
    float mn = fminf(x, y), mx = fmaxf(x, y);
    float t = mx > 0.f ? mn * p_rcp(mx) : 0.f;
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
#else
PARC_HD float p_sqrt(float x) { return sqrtf(x); }
PARC_HD float p_rcp(float x) { return 1.0f / x; }
PARC_HD void p_sincos(float x, float &s, float &c) {
    s = sinf(x);
    c = cosf(x);
}
PARC_HD float p_atan2_q1(float y, float x) { return atan2f(y, x); }
#endif

struct V3 {
    float x, y, z;
};
PARC_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PARC_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PARC_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PARC_HD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
PARC_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
PARC_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PARC_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PARC_HD V3 ld(const float *p) { return V3{p[0], p[1], p[2]}; }
PARC_HD void st(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// 3x3 row-major
struct M3 {
    float m[9];
};
PARC_HD V3 mul(const M3 &a, V3 v) {
    return V3{a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z, a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
PARC_HD V3 mulT(const M3 &a, V3 v) {
    return V3{a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z, a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z};
}
PARC_HD M3 mul(const M3 &a, const M3 &b) {
    M3 c;
    PARC_LOOP(12)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
PARC_HD M3 mulABt(const M3 &a, const M3 &b) {
    M3 c;
    PARC_LOOP(12)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[3 * j] + a.m[3 * i + 1] * b.m[3 * j + 1] + a.m[3 * i + 2] * b.m[3 * j + 2];
    return c;
}
PARC_HD M3 transpose(const M3 &a) {
    M3 c;
    PARC_LOOP(12)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * j + i];
    return c;
}
PARC_HD M3 add(const M3 &a, const M3 &b) {
    M3 c;
    PARC_LOOP(12)
    for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
    return c;
}
PARC_HD M3 sub(const M3 &a, const M3 &b) {
    M3 c;
    PARC_LOOP(12)
    for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] - b.m[i];
    return c;
}
PARC_HD M3 skew(V3 r) {
    M3 s;
    s.m[0] = 0.f; s.m[1] = -r.z; s.m[2] = r.y;
    s.m[3] = r.z; s.m[4] = 0.f; s.m[5] = -r.x;
    s.m[6] = -r.y; s.m[7] = r.x; s.m[8] = 0.f;
    return s;
}
PARC_HD M3 outer(V3 a, V3 b) {
    M3 c;
    c.m[0] = a.x * b.x; c.m[1] = a.x * b.y; c.m[2] = a.x * b.z;
    c.m[3] = a.y * b.x; c.m[4] = a.y * b.y; c.m[5] = a.y * b.z;
    c.m[6] = a.z * b.x; c.m[7] = a.z * b.y; c.m[8] = a.z * b.z;
    return c;
}
PARC_HD M3 ident(float s) {
    M3 c;
    PARC_LOOP(12)
    for (int i = 0; i < 9; ++i) c.m[i] = 0.f;
    c.m[0] = c.m[4] = c.m[8] = s;
    return c;
}
// inverse of a symmetric positive definite 3x3 (adjugate; inputs are well conditioned: inertia + augmentation)
PARC_HD M3 inv_sym(const M3 &a) {
    float a00 = a.m[0], a01 = a.m[1], a02 = a.m[2], a11 = a.m[4], a12 = a.m[5], a22 = a.m[8];
    float c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    float det = a00 * c00 + a01 * c01 + a02 * c02;
    float id = p_rcp(det);
    M3 r;
    r.m[0] = c00 * id;
    r.m[1] = r.m[3] = c01 * id;
    r.m[2] = r.m[6] = c02 * id;
    r.m[4] = (a00 * a22 - a02 * a02) * id;
    r.m[5] = r.m[7] = (a01 * a02 - a00 * a12) * id;
    r.m[8] = (a00 * a11 - a01 * a01) * id;
    return r;
}

struct Q4 {
    float x, y, z, w;
};
PARC_HD Q4 qmul(Q4 a, Q4 b) {
    return Q4{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
PARC_HD Q4 qconj(Q4 q) { return Q4{-q.x, -q.y, -q.z, q.w}; }
PARC_HD Q4 qnormalize(Q4 q) {
    float n = p_sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    float i = n > 1e-12f ? p_rcp(n) : 1.0f;
    return Q4{q.x * i, q.y * i, q.z * i, q.w * i};
}
PARC_HD M3 qmat(Q4 q) {
    float xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z, wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    M3 r;
    r.m[0] = 1.f - 2.f * (yy + zz); r.m[1] = 2.f * (xy - wz);       r.m[2] = 2.f * (xz + wy);
    r.m[3] = 2.f * (xy + wz);       r.m[4] = 1.f - 2.f * (xx + zz); r.m[5] = 2.f * (yz - wx);
    r.m[6] = 2.f * (xz - wy);       r.m[7] = 2.f * (yz + wx);       r.m[8] = 1.f - 2.f * (xx + yy);
    return r;
}
// exponential map (rotation vector) <-> quaternion; same conventions as util/torch_util.py:346-351,414-419
PARC_HD Q4 exp_to_q(V3 e) {
    float a = p_sqrt(dot(e, e));
    if (a < 1e-6f) return qnormalize(Q4{0.5f * e.x, 0.5f * e.y, 0.5f * e.z, 1.f});
    float s, c;
    p_sincos(0.5f * a, s, c);
    s *= p_rcp(a);
    return Q4{e.x * s, e.y * s, e.z * s, c};
}
PARC_HD V3 q_to_exp(Q4 q) {
    if (q.w < 0.f) q = Q4{-q.x, -q.y, -q.z, -q.w};
    float l = p_sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (l < 1e-6f) return V3{2.f * q.x, 2.f * q.y, 2.f * q.z};
    float a = 2.0f * p_atan2_q1(l, q.w) * p_rcp(l);
    return V3{a * q.x, a * q.y, a * q.z};
}

// Symmetric spatial (articulated) inertia [[A, B], [B^T, C]] acting on [angular; linear]
struct SI {
    M3 A, B, C;
};
struct SV {  // spatial motion or force vector: a = angular / moment part, l = linear / force part
    V3 a, l;
};
PARC_HD SV operator+(SV x, SV y) { return SV{x.a + y.a, x.l + y.l}; }
PARC_HD SV operator-(SV x, SV y) { return SV{x.a - y.a, x.l - y.l}; }
PARC_HD SV mul(const SI &I, SV v) { return SV{mul(I.A, v.a) + mul(I.B, v.l), mulT(I.B, v.a) + mul(I.C, v.l)}; }

// contact tuning (DESIGN.md "contact model")
struct Scratch {
    M3 E[PARC_SIM_MAX_BODIES];   // child -> parent rotation
    M3 R[PARC_SIM_MAX_BODIES];   // body -> world rotation
    V3 P[PARC_SIM_MAX_BODIES];   // body origin, world (env-local) frame
    SV v[PARC_SIM_MAX_BODIES];   // spatial velocity, body coordinates
    SV c[PARC_SIM_MAX_BODIES];   // velocity-product acceleration
    SV pA[PARC_SIM_MAX_BODIES];  // articulated bias force
    SI IA[PARC_SIM_MAX_BODIES];  // articulated inertia
    M3 Ua[PARC_SIM_MAX_BODIES], Ul[PARC_SIM_MAX_BODIES];  // U = I^A S, angular / linear rows (3 x k, k <= 3, columns used: k)
    M3 Dinv[PARC_SIM_MAX_BODIES];
    V3 u[PARC_SIM_MAX_BODIES];
    SV a[PARC_SIM_MAX_BODIES];   // spatial acceleration
    V3 flink[PARC_SIM_MAX_BODIES];   // sum of the link-link contact forces on the body this substep, body coordinates
};

// Per-env dynamic state kept in registers/scratch across the substeps of one env step
struct State {
    V3 root_pos;
    Q4 root_rot;
    SV root_vel;                        // body coordinates
    Q4 jq[PARC_SIM_MAX_BODIES];         // joint rotation (spherical / hinge as quaternion about its axis)
    float jang[PARC_SIM_MAX_BODIES];    // hinge angle
    V3 jw[PARC_SIM_MAX_BODIES];         // joint velocity: spherical omega (child frame); hinge: x = rate
    Q4 tq[PARC_SIM_MAX_BODIES];         // PD target rotation
    float tang[PARC_SIM_MAX_BODIES];    // PD target hinge angle
    V3 cforce[PARC_SIM_MAX_BODIES];     // net contact force per body, world frame, averaged over substeps
};

PARC_HD float terrain_h(const parc_terrain_t &t, int i, int j) {
    i = i < 0 ? 0 : (i > t.dim_x - 1 ? t.dim_x - 1 : i);
    j = j < 0 ? 0 : (j > t.dim_y - 1 ? t.dim_y - 1 : j);
    return t.hf[i * t.dim_y + j];
}

// One sample sphere (centre p in GLOBAL xy / env z, radius rho) against the column field, in two halves so that a caller can have the
// loads of the NEXT sphere in flight while it evaluates this one (parc_sim_bpl.h): sample_columns = cell indices + the four heights the
// evaluation needs (the own column and the three neighbours on the sphere's side, loaded together: four independent loads instead of up
// to four dependent round trips), columns_contact = the deepest contact from them: penetration depth (> 0), unit normal n.
struct ColumnSample {
    V3 p;
    int ci, cj, si, sj;
    float fx, fy;          // offset from the cell centre in cell units, [-0.5, 0.5]
    float h0, hnb[3];
};

PARC_HD ColumnSample sample_columns(const parc_terrain_t &t, V3 p) {
    ColumnSample c;
    c.p = p;
    float u = (p.x - t.min_x) * p_rcp(t.dx), w = (p.y - t.min_y) * p_rcp(t.dy);
    c.ci = (int)floorf(u + 0.5f);
    c.cj = (int)floorf(w + 0.5f);
    c.fx = u - (float)c.ci;
    c.fy = w - (float)c.cj;
    c.si = c.fx >= 0.f ? 1 : -1;
    c.sj = c.fy >= 0.f ? 1 : -1;
    c.h0 = terrain_h(t, c.ci, c.cj);
    c.hnb[0] = terrain_h(t, c.ci + c.si, c.cj);
    c.hnb[1] = terrain_h(t, c.ci, c.cj + c.sj);
    c.hnb[2] = terrain_h(t, c.ci + c.si, c.cj + c.sj);
    return c;
}

PARC_HD bool columns_contact(const parc_terrain_t &t, const ColumnSample &c, float rho, float &depth, V3 &n) {
    const V3 p = c.p;
    const int ci = c.ci, cj = c.cj, si = c.si, sj = c.sj;
    const float fx = c.fx, fy = c.fy, h0 = c.h0;
    bool hit = false;
    depth = 0.f;
    n = v3(0.f, 0.f, 1.f);
    // own column
    if (p.z - rho < h0) {
        float d_up = h0 - p.z + rho;
        float best = d_up;
        V3 bn = v3(0.f, 0.f, 1.f);
        if (d_up > 0.06f && p.z < h0) {
            // deep inside a column (walked into a wall): leave through the nearest face that opens to free space
            float dxp = (0.5f - fx) * t.dx, dxm = (0.5f + fx) * t.dx, dyp = (0.5f - fy) * t.dy, dym = (0.5f + fy) * t.dy;
            if (terrain_h(t, ci + 1, cj) < p.z && dxp + rho < best) { best = dxp + rho; bn = v3(1.f, 0.f, 0.f); }
            if (terrain_h(t, ci - 1, cj) < p.z && dxm + rho < best) { best = dxm + rho; bn = v3(-1.f, 0.f, 0.f); }
            if (terrain_h(t, ci, cj + 1) < p.z && dyp + rho < best) { best = dyp + rho; bn = v3(0.f, 1.f, 0.f); }
            if (terrain_h(t, ci, cj - 1) < p.z && dym + rho < best) { best = dym + rho; bn = v3(0.f, -1.f, 0.f); }
        }
        depth = best;
        n = bn;
        hit = true;
    }
    if (rho > 0.f) {
        // higher neighbours within reach: closest point on the neighbour's box (side face or top edge)
        PARC_LOOP(13)
        for (int k = 0; k < 3; ++k) {
            int di = (k == 1) ? 0 : si, dj = (k == 0) ? 0 : sj;
            float hn = c.hnb[k];
            if (hn <= h0 + 1e-3f) continue;
            // neighbour footprint relative to p (metres)
            float gx = di == 0 ? 0.f : ((0.5f - fabsf(fx)) * t.dx);   // distance to the shared face along x
            float gy = dj == 0 ? 0.f : ((0.5f - fabsf(fy)) * t.dy);
            float gz = p.z > hn ? p.z - hn : 0.f;
            float d2 = gx * gx + gy * gy + gz * gz;
            if (d2 >= rho * rho) continue;
            float d = p_sqrt(d2);
            float pen = rho - d;
            if (pen > depth) {
                float id = p_rcp(d);
                V3 nn = d > 1e-6f ? v3(-(float)di * gx * id, -(float)dj * gy * id, gz * id) : v3(-(float)di, -(float)dj, 0.f);
                float nl = p_sqrt(dot(nn, nn));
                if (nl > 1e-6f) {
                    n = p_rcp(nl) * nn;
                    depth = pen;
                    hit = true;
                }
            }
        }
    }
    return hit;
}

PARC_HD bool sphere_vs_columns(const parc_terrain_t &t, V3 p, float rho, float &depth, V3 &n) {
    return columns_contact(t, sample_columns(t, p), rho, depth, n);
}

PARC_HD float clampf01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }

// ---- link-link contact (self-collision) -----------------------------------------------------------------------------------
// Every body carries one capsule (model.cap_*).  Two capsules of bodies not joined by a joint that overlap push each other apart
// along the line between their closest points with a spring-damper force, plus regularised Coulomb friction.  Unlike a terrain contact
// the force is EXPLICIT and applied to the two bodies with opposite signs: an implicit one-sided impedance (what the terrain
// contact uses, the ground being immovable) would act on each link like added mass anchored in the world and let a character
// change its total momentum by rubbing its limbs together.  Both bodies evaluate the pair themselves, in canonical order (lower body
// index first), from the same closest points and gains, so the two forces are equal and opposite.  The gains are limited by the
// pair's reduced mass so that the explicit spring-damper stays inside the stability range of the semi-implicit Euler step:
// k <= mu / h^2, c <= 0.5 mu / h (mu from the two LINK masses, a lower bound of the effective masses of the articulated bodies).
struct CapsuleW {   // a body's capsule and motion in the world (env) frame
    V3 a, b;        // segment end points
    float r;
    V3 o, v, w;     // body origin, its linear velocity, angular velocity
    V3 c;           // segment mid point
    float ext;      // half length + radius: the capsule lies inside the sphere (c, ext) - cheap rejection of distant pairs
};

// closest points of two segments (clamped quadratic minimisation; degenerate segments = points are handled)
PARC_HD void seg_seg_closest(V3 p1, V3 q1, V3 p2, V3 q2, V3 &c1, V3 &c2) {
    const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    float s, t;
    const float eps = 1e-9f;
    if (a <= eps && e <= eps) {
        s = t = 0.f;
    } else if (a <= eps) {
        s = 0.f;
        t = clampf01(f * p_rcp(e));
    } else {
        const float c = dot(d1, r);
        if (e <= eps) {
            t = 0.f;
            s = clampf01(-c * p_rcp(a));
        } else {
            const float b = dot(d1, d2), den = a * e - b * b;
            s = den > eps ? clampf01((b * f - c * e) * p_rcp(den)) : 0.f;
            t = (b * s + f) * p_rcp(e);
            if (t < 0.f) {
                t = 0.f;
                s = clampf01(-c * p_rcp(a));
            } else if (t > 1.f) {
                t = 1.f;
                s = clampf01((b - c) * p_rcp(a));
            }
        }
    }
    c1 = p1 + s * d1;
    c2 = p2 + t * d2;
}

struct LinkHit {
    V3 tau, F;         // torque about this body's origin and force on this body, in the body's coordinates
};

// Contact of body A's capsule with body B's, as felt by A (R = A's rotation, mass_a / mass_b the two link masses, a_first = A has the
// lower body index).  Returns false when they do not touch or separate faster than the spring pushes.
// The pair is evaluated in CANONICAL order - the lower body index is always segment 1 of the closest-point computation and every
// quantity below is formed from (P = lower, Q = higher) - so the two lanes (or loop iterations) that own the two bodies run the same
// arithmetic on the same operands and differ only in the final sign: seg_seg_closest is not symmetric under swapping its arguments
// (near-parallel segments clamp s = 0 on segment 1), and without the ordering parallel thigh / shin capsules got two different
// closest-point pairs, i.e. forces that were not equal and opposite.
// Friction between links (Isaac Gym: collision filter 0 = link-link contacts with the material's friction, envs/ig_char_env.py:105-113):
// regularised Coulomb, |Ft| = min(ct |vt|, mu fn) against the tangential relative velocity, applied to both bodies at the COMMON point
// midway between the two surface points (equal and opposite forces on one line of action: no net force, no net torque); the normal
// forces act at the two surface points, which lie on one line along n.
PARC_HD bool link_contact(const parc_sim_model_t &m, const CapsuleW &A, const M3 &R, float mass_a, const CapsuleW &B, float mass_b, float h,
                          bool a_first, LinkHit &hit) {
    {
        const V3 cc = A.c - B.c;
        const float far = A.ext + B.ext;
        if (dot(cc, cc) >= far * far) return false;        // bounding spheres apart (most pairs, most of the time)
    }
    const CapsuleW &P = a_first ? A : B, &Q = a_first ? B : A;
    V3 cp, cq;
    seg_seg_closest(P.a, P.b, Q.a, Q.b, cp, cq);
    V3 d = cp - cq;
    const float dist2 = dot(d, d), reach = P.r + Q.r;
    if (dist2 >= reach * reach) return false;
    const float dist = p_sqrt(dist2);
    // coincident axes: push along the line between the body origins
    V3 n = dist > 1e-6f ? p_rcp(dist) * d : v3(0.f, 0.f, 0.f);          // from Q towards P
    if (!(dist > 1e-6f)) {
        V3 oo = P.o - Q.o;
        float l = p_sqrt(dot(oo, oo));
        n = l > 1e-6f ? p_rcp(l) * oo : v3(0.f, 0.f, 1.f);
    }
    const float pen = reach - dist;
    const V3 pp = cp - P.r * n, pq = cq + Q.r * n;            // surface points
    const V3 vp = P.v + cross(P.w, pp - P.o), vq = Q.v + cross(Q.w, pq - Q.o);
    const float vn = dot(vp - vq, n);
    const float mu = mass_a * mass_b * p_rcp(mass_a + mass_b);
    const float ih = p_rcp(h);
    const float kcap = mu * ih * ih, ccap = 0.5f * mu * ih;
    const float kn = m.contact_kn < kcap ? m.contact_kn : kcap;
    const float cn = vn < 0.f ? (m.contact_cn < ccap ? m.contact_cn : ccap) : 0.f;
    const float d_eff = pen < m.contact_max_pen ? pen : m.contact_max_pen;
    const float fn = kn * d_eff - cn * vn;
    if (fn <= 0.f) return false;
    // friction on P at the common point (explicit damper, gain inside the step's stability range like the normal one, capped by mu fn)
    const V3 pm = 0.5f * (pp + pq);
    const V3 vr = (P.v + cross(P.w, pm - P.o)) - (Q.v + cross(Q.w, pm - Q.o));
    const V3 vt = vr - dot(vr, n) * n;
    const float vt2 = dot(vt, vt);
    V3 Ft = v3(0.f, 0.f, 0.f);
    if (vt2 > 1e-12f) {
        const float vtl = p_sqrt(vt2);
        const float ctc = 0.5f * ccap;
        const float ct = m.contact_ct < ctc ? m.contact_ct : ctc;
        const float lim = m.friction_mu * fn;
        const float ft = ct * vtl < lim ? ct * vtl : lim;
        Ft = (-ft * p_rcp(vtl)) * vt;
    }
    const float sg = a_first ? 1.f : -1.f;
    const V3 Fn = (sg * fn) * n;
    Ft = sg * Ft;
    const V3 pa = a_first ? pp : pq;
    hit.tau = mulT(R, cross(pa - A.o, Fn) + cross(pm - A.o, Ft));
    hit.F = mulT(R, Fn + Ft);
    return true;
}

PARC_HD CapsuleW capsule_world(const parc_sim_model_t &m, int b, const M3 &R, V3 P, SV v_body) {
    CapsuleW c;
    c.a = P + mul(R, ld(m.cap_p0[b]));
    c.b = P + mul(R, ld(m.cap_p1[b]));
    c.r = m.cap_radius[b];
    c.o = P;
    c.v = mul(R, v_body.l);
    c.w = mul(R, v_body.a);
    c.c = 0.5f * (c.a + c.b);
    const V3 ax = c.b - c.a;
    c.ext = 0.5f * p_sqrt(dot(ax, ax)) + c.r;
    return c;
}

// Forward kinematics + velocities + bias terms (ABA pass 1) and per-body contact impedance.
// After this call s.IA / s.pA hold the rigid-body inertia + contact augmentation and the bias force minus
// external forces; contact bookkeeping needed to report forces afterwards is recomputed in report_contacts().
PARC_HD void pass1(const parc_sim_model_t &m, const parc_terrain_t &ter, V3 env_off, const State &x, Scratch &s, float h) {
    const int B = m.num_bodies;
    s.R[0] = qmat(x.root_rot);
    s.P[0] = x.root_pos;
    s.v[0] = x.root_vel;
    s.c[0] = SV{v3(0, 0, 0), v3(0, 0, 0)};
    PARC_LOOP(0)
    for (int i = 1; i < B; ++i) {
        const int p = m.parent[i];
        M3 E = mul(qmat(Q4{m.local_rotation[i][0], m.local_rotation[i][1], m.local_rotation[i][2], m.local_rotation[i][3]}), qmat(x.jq[i]));
        s.E[i] = E;
        V3 r = ld(m.local_translation[i]);
        s.R[i] = mul(s.R[p], E);
        s.P[i] = s.P[p] + mul(s.R[p], r);
        SV vp = s.v[p];
        V3 wj = v3(0, 0, 0);
        if (m.joint_type[i] == PARC_JOINT_SPHERICAL) wj = x.jw[i];
        else if (m.joint_type[i] == PARC_JOINT_HINGE) wj = x.jw[i].x * ld(m.joint_axis[i]);
        SV vi;
        vi.a = mulT(E, vp.a) + wj;
        vi.l = mulT(E, vp.l + cross(vp.a, r));
        s.v[i] = vi;
        s.c[i] = SV{cross(vi.a, wj), cross(vi.l, wj)};
    }
    PARC_LOOP(1)
    for (int i = 0; i < B; ++i) {
        const float mass = m.mass[i];
        V3 hc = mass * ld(m.com[i]);
        SI I;
        const float *io = m.inertia_o[i];
        I.A.m[0] = io[0]; I.A.m[1] = io[1]; I.A.m[2] = io[2];
        I.A.m[3] = io[1]; I.A.m[4] = io[3]; I.A.m[5] = io[4];
        I.A.m[6] = io[2]; I.A.m[7] = io[4]; I.A.m[8] = io[5];
        I.B = skew(hc);
        I.C = ident(mass);
        SV v = s.v[i];
        SV Iv = mul(I, v);
        // p = v x* (I v) - f_ext ;  [w;v] x* [n;f] = [w x n + v x f ; w x f]
        SV p;
        p.a = cross(v.a, Iv.a) + cross(v.l, Iv.l);
        p.l = cross(v.a, Iv.l);
        V3 fg = mulT(s.R[i], v3(0.f, 0.f, -m.gravity * mass));   // gravity at the centre of mass
        p.a = p.a - cross(ld(m.com[i]), fg);
        p.l = p.l - fg;
        // per-link angular damping: the couple -c I_com w, with I_com w = I_o w + m c x (c x w)
        V3 cm = ld(m.com[i]);
        p.a = p.a + m.angular_damping * (mul(I.A, v.a) + mass * cross(cm, cross(cm, v.a)));
        s.IA[i] = I;
        s.pA[i] = p;
    }
    // link-link contacts: explicit, equal and opposite (see link_contact)
    for (int i = 0; i < B; ++i) {
        s.flink[i] = v3(0.f, 0.f, 0.f);
        if (!(m.cap_radius[i] > 0.f) || m.self_mask[i] == 0u) continue;
        const CapsuleW ci = capsule_world(m, i, s.R[i], s.P[i], s.v[i]);
        for (int j = 0; j < B; ++j) {
            if (!((m.self_mask[i] >> j) & 1u) || !(m.cap_radius[j] > 0.f)) continue;
            LinkHit hit;
            if (!link_contact(m, ci, s.R[i], m.mass[i], capsule_world(m, j, s.R[j], s.P[j], s.v[j]), m.mass[j], h, i < j, hit)) continue;
            s.pA[i].a = s.pA[i].a - hit.tau;
            s.pA[i].l = s.pA[i].l - hit.F;
            s.flink[i] = s.flink[i] + hit.F;
        }
    }
    // contacts: implicit spring-damper + regularised friction per penetrating sample sphere
    PARC_LOOP(2)
    for (int k = 0; k < m.num_spheres; ++k) {
        const int b = m.sph_body[k];
        V3 rb = ld(m.sph_pos[k]);
        V3 pw = s.P[b] + mul(s.R[b], rb);
        float depth;
        V3 n;
        if (!sphere_vs_columns(ter, pw + env_off, m.sph_radius[k], depth, n)) continue;
        V3 rc = rb - m.sph_radius[k] * mulT(s.R[b], n);      // contact point, body coordinates
        V3 vpb = s.v[b].l + cross(s.v[b].a, rc);             // its velocity, body coordinates
        V3 nb = mulT(s.R[b], n);
        float vn = dot(vpb, nb);
        float d_eff = depth < m.contact_max_pen ? depth : m.contact_max_pen;
        float cn = m.contact_cn;      // damper on approach AND on rebound (restitution 0, envs/ig_env.py:517,733); fn0 <= 0 below = no adhesion
        float fn0 = m.contact_kn * d_eff - cn * vn;
        if (fn0 <= 0.f) continue;
        V3 vt = vpb - vn * nb;
        float vtn = p_sqrt(dot(vt, vt));
        float ct = m.contact_ct;
        float ct_cone = m.friction_mu * fn0 * p_rcp(vtn > 1e-4f ? vtn : 1e-4f);
        if (ct_cone < ct) ct = ct_cone;
        V3 F0 = fn0 * nb - ct * vt;
        // The impedance Z acts on the change of the contact point's WORLD velocity over the step.  The contact point of a sample sphere is
        // c - rad n with c the sphere's centre (a material point) and n fixed in the world, so that change is
        // h R (a_sp.l + alpha x rc + w x v_c), v_c = v + w x rb the centre's velocity: the first two terms are the unknown the impedance
        // multiplies, the third (the body frame turns by h w during the step) is known and belongs to the explicit force - without it a
        // spinning body's contact predicts an approach speed -h (w x v_c).n that is not there and leaves the friction cone.
        {
            V3 wv = h * cross(s.v[b].a, s.v[b].l + cross(s.v[b].a, rb));
            float wvn = dot(wv, nb);
            F0 = F0 - ((cn + h * m.contact_kn) * wvn) * nb - ct * (wv - wvn * nb);
        }
        // Z = (cn + h kn) n n^T + ct (1 - n n^T), body coordinates
        M3 Z = add(ident(ct), outer(((cn + h * m.contact_kn) - ct) * nb, nb));
        M3 Sr = skew(rc);
        M3 SZ = mul(Sr, Z);
        // A += h * (-Sr Z Sr), B += h * (Sr Z), C += h * Z
        M3 SZS = mul(SZ, Sr);
        PARC_LOOP(12)
        for (int q = 0; q < 9; ++q) {
            s.IA[b].A.m[q] -= h * SZS.m[q];
            s.IA[b].B.m[q] += h * SZ.m[q];
            s.IA[b].C.m[q] += h * Z.m[q];
        }
        s.pA[b].a = s.pA[b].a - cross(rc, F0);
        s.pA[b].l = s.pA[b].l - F0;
    }
}

// After the accelerations are known: realised contact force of every body, world frame (F+ = F0 - Z h J a)
PARC_HD void report_contacts(const parc_sim_model_t &m, const parc_terrain_t &ter, V3 env_off, const Scratch &s, float h, float weight, State &x) {
    PARC_LOOP(3)
    for (int k = 0; k < m.num_spheres; ++k) {
        const int b = m.sph_body[k];
        V3 rb = ld(m.sph_pos[k]);
        V3 pw = s.P[b] + mul(s.R[b], rb);
        float depth;
        V3 n;
        if (!sphere_vs_columns(ter, pw + env_off, m.sph_radius[k], depth, n)) continue;
        V3 rc = rb - m.sph_radius[k] * mulT(s.R[b], n);
        V3 vpb = s.v[b].l + cross(s.v[b].a, rc);
        V3 nb = mulT(s.R[b], n);
        float vn = dot(vpb, nb);
        float d_eff = depth < m.contact_max_pen ? depth : m.contact_max_pen;
        float cn = m.contact_cn;      // damper on approach AND on rebound (restitution 0, envs/ig_env.py:517,733); fn0 <= 0 below = no adhesion
        float fn0 = m.contact_kn * d_eff - cn * vn;
        if (fn0 <= 0.f) continue;
        V3 vt = vpb - vn * nb;
        float vtn = p_sqrt(dot(vt, vt));
        float ct = m.contact_ct;
        float ct_cone = m.friction_mu * fn0 * p_rcp(vtn > 1e-4f ? vtn : 1e-4f);
        if (ct_cone < ct) ct = ct_cone;
        V3 F0 = fn0 * nb - ct * vt;
        V3 dv = h * (s.a[b].l + cross(s.a[b].a, rc) + cross(s.v[b].a, s.v[b].l + cross(s.v[b].a, rb)));   // world-velocity change of the contact point (pass1)
        float dvn = dot(dv, nb);
        V3 F = F0 - (cn + h * m.contact_kn) * dvn * nb - ct * (dv - dvn * nb);
        float fnn = dot(F, nb);
        if (fnn < 0.f) F = F - fnn * nb;   // no adhesion in what is reported
        x.cforce[b] = x.cforce[b] + weight * mul(s.R[b], F);
    }
    // link-link contacts count in the net contact force of a body like any other contact (Isaac Gym's net_contact_force tensor)
    for (int i = 0; i < m.num_bodies; ++i) x.cforce[i] = x.cforce[i] + weight * mul(s.R[i], s.flink[i]);
}

PARC_HD float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ABA passes 2 and 3 + integration: one substep of length h.
PARC_HD void substep(const parc_sim_model_t &m, const parc_terrain_t &ter, V3 env_off, State &x, Scratch &s, float h, float cweight) {
    const int B = m.num_bodies;
    pass1(m, ter, env_off, x, s, h);
    // ---- joint drives (implicit PD) and limits: tau and the diagonal augmentation of D
    // ---- pass 2: leaves -> root
    PARC_LOOP(4)
    for (int i = B - 1; i >= 1; --i) {
        const int p = m.parent[i];
        const int jt = m.joint_type[i];
        const int d0 = m.dof_idx[i];
        SI Ia = s.IA[i];
        SV pa = s.pA[i];
        if (jt == PARC_JOINT_SPHERICAL) {
            // predicted rotation error in the child frame: log(q^-1 q_target) - h w
            V3 err = q_to_exp(qmul(qconj(x.jq[i]), x.tq[i])) - h * x.jw[i];
            V3 e = q_to_exp(x.jq[i]);
            float tau[3], aug[3];
            const float ev[3] = {err.x, err.y, err.z}, wv[3] = {x.jw[i].x, x.jw[i].y, x.jw[i].z}, ee[3] = {e.x, e.y, e.z};
            PARC_LOOP(5)
            for (int k = 0; k < 3; ++k) {
                float kp = m.kp[d0 + k], kd = m.kd[d0 + k];
                float t = kp * ev[k] - kd * wv[k];
                float lim = m.effort[d0 + k];
                float sc = (lim > 0.f && fabsf(t) > lim) ? lim * p_rcp(fabsf(t)) : 1.0f;
                tau[k] = sc * t;
                aug[k] = m.armature[d0 + k] + sc * (h * kd + h * h * kp);
                float over = ee[k] > m.limit_hi[d0 + k] ? ee[k] - m.limit_hi[d0 + k] : (ee[k] < m.limit_lo[d0 + k] ? ee[k] - m.limit_lo[d0 + k] : 0.f);
                if (over != 0.f) {
                    tau[k] += -m.limit_kp * (over + h * wv[k]) - m.limit_kd * wv[k];
                    aug[k] += h * m.limit_kd + h * h * m.limit_kp;
                }
            }
            M3 D = Ia.A;
            D.m[0] += aug[0]; D.m[4] += aug[1]; D.m[8] += aug[2];
            M3 Di = inv_sym(D);
            M3 Ua = Ia.A;             // U = I^A S with S = [1;0]:  Ua = A, Ul = B^T
            M3 Ul = transpose(Ia.B);
            V3 u = v3(tau[0], tau[1], tau[2]) - pa.a;
            s.Ua[i] = Ua; s.Ul[i] = Ul; s.Dinv[i] = Di; s.u[i] = u;
            // Ia = I^A - U Dinv U^T
            M3 UaDi = mul(Ua, Di), UlDi = mul(Ul, Di);
            Ia.A = sub(Ia.A, mulABt(UaDi, Ua));
            Ia.B = sub(Ia.B, mulABt(UaDi, Ul));
            Ia.C = sub(Ia.C, mulABt(UlDi, Ul));
            SV Iac = mul(Ia, s.c[i]);
            V3 Diu = mul(Di, u);
            pa.a = pa.a + Iac.a + mul(Ua, Diu);
            pa.l = pa.l + Iac.l + mul(Ul, Diu);
        } else if (jt == PARC_JOINT_HINGE) {
            V3 ax = ld(m.joint_axis[i]);
            float w = x.jw[i].x;
            float kp = m.kp[d0], kd = m.kd[d0];
            float t = kp * (x.tang[i] - x.jang[i] - h * w) - kd * w;
            float lim = m.effort[d0];
            float sc = (lim > 0.f && fabsf(t) > lim) ? lim * p_rcp(fabsf(t)) : 1.0f;
            float tau = sc * t;
            float aug = m.armature[d0] + sc * (h * kd + h * h * kp);
            float ang = x.jang[i];
            float over = ang > m.limit_hi[d0] ? ang - m.limit_hi[d0] : (ang < m.limit_lo[d0] ? ang - m.limit_lo[d0] : 0.f);
            if (over != 0.f) {
                tau += -m.limit_kp * (over + h * w) - m.limit_kd * w;
                aug += h * m.limit_kd + h * h * m.limit_kp;
            }
            V3 ua = mul(Ia.A, ax), ul = mulT(Ia.B, ax);
            float D = dot(ax, ua) + aug;
            float Di = p_rcp(D);
            float u = tau - dot(ax, pa.a);
            s.Ua[i].m[0] = ua.x; s.Ua[i].m[1] = ua.y; s.Ua[i].m[2] = ua.z;
            s.Ul[i].m[0] = ul.x; s.Ul[i].m[1] = ul.y; s.Ul[i].m[2] = ul.z;
            s.Dinv[i].m[0] = Di;
            s.u[i] = v3(u, 0.f, 0.f);
            Ia.A = sub(Ia.A, outer(Di * ua, ua));
            Ia.B = sub(Ia.B, outer(Di * ua, ul));
            Ia.C = sub(Ia.C, outer(Di * ul, ul));
            SV Iac = mul(Ia, s.c[i]);
            pa.a = pa.a + Iac.a + (Di * u) * ua;
            pa.l = pa.l + Iac.l + (Di * u) * ul;
        }
        // fixed joint: the whole articulated inertia passes to the parent (c = 0)
        // transform to the parent frame: rotate by E, shift by r
        const M3 &E = s.E[i];
        V3 r = ld(m.local_translation[i]);
        M3 Ar = mulABt(mul(E, Ia.A), E), Br = mulABt(mul(E, Ia.B), E), Cr = mulABt(mul(E, Ia.C), E);
        M3 S = skew(r);
        M3 SC = mul(S, Cr);
        M3 SBt = mulABt(S, Br);        // S B^T
        // A'' = A + S B^T + (S B^T)^T - S C S ; B'' = B + S C
        M3 SCS = mul(SC, S);
        M3 Ap = sub(add(add(Ar, SBt), transpose(SBt)), SCS);
        M3 Bp = add(Br, SC);
        SI &Ip = s.IA[p];
        Ip.A = add(Ip.A, Ap);
        Ip.B = add(Ip.B, Bp);
        Ip.C = add(Ip.C, Cr);
        V3 fl = mul(E, pa.l);
        V3 fa = mul(E, pa.a) + cross(r, fl);
        s.pA[p].a = s.pA[p].a + fa;
        s.pA[p].l = s.pA[p].l + fl;
    }
    // ---- floating base: [[A,B],[B^T,C]] [alpha; a] = -[pn; pf]
    {
        const SI &I0 = s.IA[0];
        M3 Ci = inv_sym(I0.C);
        M3 BCi = mul(I0.B, Ci);
        M3 Sc = sub(I0.A, mulABt(BCi, I0.B));   // A - B C^-1 B^T
        M3 Sci = inv_sym(Sc);
        V3 pn = s.pA[0].a, pf = s.pA[0].l;
        V3 alpha = mul(Sci, mul(BCi, pf) - pn);
        V3 lin = mul(Ci, -(pf + mulT(I0.B, alpha)));
        s.a[0] = SV{alpha, lin};
    }
    // ---- pass 3: root -> leaves, joint accelerations and velocity update
    PARC_LOOP(6)
    for (int i = 1; i < B; ++i) {
        const int p = m.parent[i];
        const int jt = m.joint_type[i];
        const M3 &E = s.E[i];
        V3 r = ld(m.local_translation[i]);
        SV ap = s.a[p];
        SV a1;
        a1.a = mulT(E, ap.a) + s.c[i].a;
        a1.l = mulT(E, ap.l + cross(ap.a, r)) + s.c[i].l;
        if (jt == PARC_JOINT_SPHERICAL) {
            V3 rhs = s.u[i] - (mulT(s.Ua[i], a1.a) + mulT(s.Ul[i], a1.l));
            V3 qdd = mul(s.Dinv[i], rhs);
            a1.a = a1.a + qdd;
            x.jw[i] = x.jw[i] + h * qdd;
        } else if (jt == PARC_JOINT_HINGE) {
            V3 ua = v3(s.Ua[i].m[0], s.Ua[i].m[1], s.Ua[i].m[2]), ul = v3(s.Ul[i].m[0], s.Ul[i].m[1], s.Ul[i].m[2]);
            float qdd = s.Dinv[i].m[0] * (s.u[i].x - dot(ua, a1.a) - dot(ul, a1.l));
            a1.a = a1.a + qdd * ld(m.joint_axis[i]);
            x.jw[i].x += h * qdd;
        }
        s.a[i] = a1;
    }
    if (cweight > 0.f) report_contacts(m, ter, env_off, s, h, cweight, x);
    // ---- integrate (semi-implicit Euler): velocities first, then positions with the new velocities
    // The root's linear velocity advances in WORLD coordinates: v_w += h R (a_sp + w x v) (the classical acceleration of the origin).
    // Advancing its body coordinates by h a_sp instead turns them by (1 - h [w]), which stretches |v| by 1 + h^2 w^2 / 2 per substep (16 %
    // per second at 6 rad/s and h = 1/120: linear momentum of a tumbling character was not conserved - round 4, spinning-ball test).
    const V3 vw_new = mul(s.R[0], x.root_vel.l + h * (s.a[0].l + cross(x.root_vel.a, x.root_vel.l)));
    x.root_vel.a = x.root_vel.a + h * s.a[0].a;
    const float wmax = m.max_angular_velocity;
    {
        float wn = p_sqrt(dot(x.root_vel.a, x.root_vel.a));
        if (wn > wmax) x.root_vel.a = (wmax * p_rcp(wn)) * x.root_vel.a;
    }
    x.root_rot = qnormalize(qmul(x.root_rot, exp_to_q(h * x.root_vel.a)));
    x.root_pos = x.root_pos + h * vw_new;
    x.root_vel.l = mulT(qmat(x.root_rot), vw_new);         // back to the coordinates of the new body frame
    PARC_LOOP(7)
    for (int i = 1; i < B; ++i) {
        const int jt = m.joint_type[i];
        if (jt == PARC_JOINT_SPHERICAL) {
            float wn = p_sqrt(dot(x.jw[i], x.jw[i]));
            if (wn > wmax) x.jw[i] = (wmax * p_rcp(wn)) * x.jw[i];
            x.jq[i] = qnormalize(qmul(x.jq[i], exp_to_q(h * x.jw[i])));
        } else if (jt == PARC_JOINT_HINGE) {
            x.jw[i].x = clampf(x.jw[i].x, -wmax, wmax);
            x.jang[i] += h * x.jw[i].x;
            V3 ax = ld(m.joint_axis[i]);
            x.jq[i] = exp_to_q(x.jang[i] * ax);
        }
    }
}

// ---- state <-> Isaac-Gym-layout tensors (envs/ig_env.py:764-780) -------------------------------------------
PARC_HD void load_state(const parc_sim_model_t &m, const float *root_state, const float *dof_state, const float *action,
                        const float *act_lo, const float *act_hi, State &x) {
    x.root_pos = ld(root_state);
    x.root_rot = qnormalize(Q4{root_state[3], root_state[4], root_state[5], root_state[6]});
    M3 R = qmat(x.root_rot);
    x.root_vel.l = mulT(R, ld(root_state + 7));    // world -> body coordinates
    x.root_vel.a = mulT(R, ld(root_state + 10));
    PARC_LOOP(8)
    for (int i = 0; i < m.num_bodies; ++i) {
        x.jq[i] = Q4{0.f, 0.f, 0.f, 1.f};
        x.tq[i] = Q4{0.f, 0.f, 0.f, 1.f};
        x.jang[i] = 0.f;
        x.tang[i] = 0.f;
        x.jw[i] = v3(0, 0, 0);
        x.cforce[i] = v3(0, 0, 0);
        const int d0 = m.dof_idx[i];
        if (m.joint_type[i] == PARC_JOINT_SPHERICAL) {
            x.jq[i] = exp_to_q(v3(dof_state[2 * d0], dof_state[2 * (d0 + 1)], dof_state[2 * (d0 + 2)]));
            x.jw[i] = v3(dof_state[2 * d0 + 1], dof_state[2 * (d0 + 1) + 1], dof_state[2 * (d0 + 2) + 1]);
            float t[3];
            PARC_LOOP(8)
            for (int k = 0; k < 3; ++k) t[k] = clampf(action[d0 + k], act_lo[d0 + k], act_hi[d0 + k]);   // ig_char_env.py:490
            x.tq[i] = exp_to_q(v3(t[0], t[1], t[2]));
        } else if (m.joint_type[i] == PARC_JOINT_HINGE) {
            x.jang[i] = dof_state[2 * d0];
            x.jw[i].x = dof_state[2 * d0 + 1];
            x.jq[i] = exp_to_q(x.jang[i] * ld(m.joint_axis[i]));
            x.tang[i] = clampf(action[d0], act_lo[d0], act_hi[d0]);
        }
    }
}

PARC_HD void store_state(const parc_sim_model_t &m, const State &x, const Scratch &s, float *root_state, float *dof_state,
                         float *rigid_body_state, float *contact_forces) {
    M3 R = qmat(x.root_rot);
    st(root_state, x.root_pos);
    root_state[3] = x.root_rot.x; root_state[4] = x.root_rot.y; root_state[5] = x.root_rot.z; root_state[6] = x.root_rot.w;
    st(root_state + 7, mul(R, x.root_vel.l));
    st(root_state + 10, mul(R, x.root_vel.a));
    PARC_LOOP(9)
    for (int i = 1; i < m.num_bodies; ++i) {
        const int d0 = m.dof_idx[i];
        if (m.joint_type[i] == PARC_JOINT_SPHERICAL) {
            V3 e = q_to_exp(x.jq[i]);
            dof_state[2 * d0] = e.x; dof_state[2 * (d0 + 1)] = e.y; dof_state[2 * (d0 + 2)] = e.z;
            dof_state[2 * d0 + 1] = x.jw[i].x; dof_state[2 * (d0 + 1) + 1] = x.jw[i].y; dof_state[2 * (d0 + 2) + 1] = x.jw[i].z;
        } else if (m.joint_type[i] == PARC_JOINT_HINGE) {
            dof_state[2 * d0] = x.jang[i];
            dof_state[2 * d0 + 1] = x.jw[i].x;
        }
    }
    (void)s;
    (void)rigid_body_state;
    (void)contact_forces;
}

// Body poses / velocities of the CURRENT state (after the last substep) -> rigid_body_state [B,13], contact forces [B,3]
PARC_HD void publish_bodies(const parc_sim_model_t &m, const State &x, float *rigid_body_state, float *contact_forces) {
    M3 R[PARC_SIM_MAX_BODIES];
    Q4 Q[PARC_SIM_MAX_BODIES];
    V3 P[PARC_SIM_MAX_BODIES];
    SV v[PARC_SIM_MAX_BODIES];
    R[0] = qmat(x.root_rot);
    Q[0] = x.root_rot;
    P[0] = x.root_pos;
    v[0] = x.root_vel;
    PARC_LOOP(10)
    for (int i = 1; i < m.num_bodies; ++i) {
        const int p = m.parent[i];
        Q4 lq = qmul(Q4{m.local_rotation[i][0], m.local_rotation[i][1], m.local_rotation[i][2], m.local_rotation[i][3]}, x.jq[i]);
        M3 E = qmat(lq);
        V3 r = ld(m.local_translation[i]);
        Q[i] = qnormalize(qmul(Q[p], lq));
        R[i] = mul(R[p], E);
        P[i] = P[p] + mul(R[p], r);
        V3 wj = v3(0, 0, 0);
        if (m.joint_type[i] == PARC_JOINT_SPHERICAL) wj = x.jw[i];
        else if (m.joint_type[i] == PARC_JOINT_HINGE) wj = x.jw[i].x * ld(m.joint_axis[i]);
        v[i].a = mulT(E, v[p].a) + wj;
        v[i].l = mulT(E, v[p].l + cross(v[p].a, r));
    }
    PARC_LOOP(10)
    for (int i = 0; i < m.num_bodies; ++i) {
        float *o = rigid_body_state + 13 * i;
        st(o, P[i]);
        o[3] = Q[i].x; o[4] = Q[i].y; o[5] = Q[i].z; o[6] = Q[i].w;
        st(o + 7, mul(R[i], v[i].l));
        st(o + 10, mul(R[i], v[i].a));
        st(contact_forces + 3 * i, x.cforce[i]);
    }
}

// One env step: `n_sub` substeps of length h with the PD targets held (envs/ig_env.py:830-837: sim_steps x substeps)
PARC_HD void env_step(const parc_sim_model_t &m, const parc_terrain_t &ter, const float *env_offset, float *root_state, float *dof_state,
                      float *rigid_body_state, float *contact_forces, const float *action, const float *act_lo, const float *act_hi,
                      int n_sub, float h, Scratch &s) {
    State x;
#if defined(PARC_SIM_FILL_STATE)     // host test builds: start from a known byte pattern (oracle/Makefile `poison`)
    memset((void *)&x, PARC_SIM_FILL_STATE, sizeof x);
#endif
    load_state(m, root_state, dof_state, action, act_lo, act_hi, x);
    V3 off = ld(env_offset);
    const float w = 1.0f / (float)n_sub;
    PARC_LOOP(11)
    for (int k = 0; k < n_sub; ++k) substep(m, ter, off, x, s, h, w);
    store_state(m, x, s, root_state, dof_state, rigid_body_state, contact_forces);
    publish_bodies(m, x, rigid_body_state, contact_forces);
}

}  // namespace parc_sim
