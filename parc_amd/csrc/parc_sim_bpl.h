// Articulated-body step, BODY PER LANE: 16 lanes = one env (lane b = body b), 4 envs per wave.
//
// Same equations as parc_sim_core.h (which stays the single-source host/device reference: one env per lane, serial
// loops over the bodies).  At 4096 envs that layout is only 64 wavefronts on a 1024-SIMD chip, each walking a
// ~100k-instruction dependent stream per substep.  Here the three ABA sweeps are level-synchronous over the kinematic
// tree (depth 4 for the humanoid): a sweep costs `depth` steps instead of `bodies` steps, a 4096-env launch is 1024
// wavefronts, and every per-body quantity (6x6 articulated inertia, U, D^-1, velocities) lives in that lane's registers
// instead of per-thread scratch arrays.  Parent <- child accumulation of the inward sweep goes through LDS (27 floats per
// child), parent -> child broadcasts of the outward sweeps are 16-lane shuffles.
#pragma once
#if !defined(PARC_LANE_EMU)
#include <hip/hip_runtime.h>
#endif
// (PARC_LANE_EMU: oracle/sim_host_bpl.cpp compiles this very header for the host, with the lane primitives __shfl / __shfl_xor /
// __ballot / __syncthreads / threadIdx provided by a 16-fiber lock-step emulation, so that sanitizers and the CPU invariant tests
// reach the kernel the product runs -- test infrastructure, never linked into libparc_hip.so)

#include "parc_sim_core.h"

namespace parc_sim_bpl {
using namespace parc_sim;

#define BPL_G 16           // lanes per env
#define BPL_EPB 4          // envs per 64-thread workgroup
#define BPL_CONTRIB 27     // A(6 sym) + B(9) + C(6 sym) + moment(3) + force(3)

__device__ __forceinline__ float shf(float v, int src) { return __shfl(v, src, BPL_G); }
__device__ __forceinline__ V3 shf(V3 v, int src) { return V3{shf(v.x, src), shf(v.y, src), shf(v.z, src)}; }
__device__ __forceinline__ SV shf(SV v, int src) { return SV{shf(v.a, src), shf(v.l, src)}; }
__device__ __forceinline__ M3 shf(const M3 &a, int src) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = shf(a.m[i], src);
    return r;
}

// per-lane constants of body b
struct Lane {
    int p, jt, d0, depth;
    V3 r, ax, com;
    M3 El;                 // rotation of the joint frame in the parent (local_rotation)
    Q4 lrot;               // the same as a quaternion
    float mass, io[6];
    unsigned long long sph;   // bit k: sample sphere k belongs to this body
    unsigned children;        // bit c: body c is a child
};

__device__ __forceinline__ Lane load_lane(const parc_sim_model_t &m, int b) {
    Lane L;
    const bool valid = b < m.num_bodies;
    const int bb = valid ? b : 0;
    L.p = (valid && b > 0) ? m.parent[bb] : 0;
    L.jt = valid ? m.joint_type[bb] : -1;
    L.d0 = m.dof_idx[bb];
    L.r = ld(m.local_translation[bb]);
    L.ax = ld(m.joint_axis[bb]);
    L.com = ld(m.com[bb]);
    L.lrot = Q4{m.local_rotation[bb][0], m.local_rotation[bb][1], m.local_rotation[bb][2], m.local_rotation[bb][3]};
    L.El = qmat(L.lrot);
    L.mass = m.mass[bb];
#pragma unroll
    for (int i = 0; i < 6; ++i) L.io[i] = m.inertia_o[bb][i];
    int d = 0;
    for (int a = bb; a > 0 && d < PARC_SIM_MAX_BODIES; a = m.parent[a]) ++d;
    L.depth = valid ? d : -1;
    L.sph = 0ull;
    for (int k = 0; k < m.num_spheres; ++k)
        if (valid && m.sph_body[k] == b) L.sph |= 1ull << k;
    L.children = 0u;
    for (int c = 1; c < m.num_bodies; ++c)
        if (valid && m.parent[c] == b) L.children |= 1u << c;
    return L;
}

// per-lane dynamic state of body b (lane 0 carries the floating base in root_*)
struct LState {
    V3 root_pos;
    Q4 root_rot;
    SV root_vel;      // body coordinates
    Q4 jq, tq;
    float jang, tang;
    V3 jw;
    V3 cforce;
};

struct Kin {
    M3 E, R;
    V3 P;
    SV v, c;
};

// outward sweep 1: frames and velocities, level-synchronous
__device__ __forceinline__ void kin_pass(const Lane &L, int b, int maxd, const LState &x, Kin &k) {
    V3 wj = v3(0, 0, 0);
    if (L.jt == PARC_JOINT_SPHERICAL) wj = x.jw;
    else if (L.jt == PARC_JOINT_HINGE) wj = x.jw.x * L.ax;
    k.E = mul(L.El, qmat(x.jq));
    k.R = qmat(x.root_rot);          // meaningful on lane 0, overwritten elsewhere
    k.P = x.root_pos;
    k.v = x.root_vel;
    k.c = SV{v3(0, 0, 0), v3(0, 0, 0)};
    for (int l = 1; l <= maxd; ++l) {
        M3 Rp = shf(k.R, L.p);
        V3 Pp = shf(k.P, L.p);
        SV vp = shf(k.v, L.p);
        if (L.depth == l) {
            k.R = mul(Rp, k.E);
            k.P = Pp + mul(Rp, L.r);
            k.v.a = mulT(k.E, vp.a) + wj;
            k.v.l = mulT(k.E, vp.l + cross(vp.a, L.r));
            k.c = SV{cross(k.v.a, wj), cross(k.v.l, wj)};
        }
    }
}

struct ContactEval {
    bool hit;
    V3 rc, nb, F0;
    float cn, ct;
};

// one sample sphere of this lane's body: the same arithmetic as pass1 / report_contacts of the reference core
// the heightfield cells under sample sphere s of this lane's body (loads only)
__device__ __forceinline__ ColumnSample sample_sphere(const parc_sim_model_t &m, const parc_terrain_t &ter, V3 env_off, const Kin &k, int s) {
    return sample_columns(ter, k.P + mul(k.R, ld(m.sph_pos[s])) + env_off);
}

__device__ __forceinline__ ContactEval eval_contact(const parc_sim_model_t &m, const parc_terrain_t &ter, const ColumnSample &cs, const Kin &k, int s, float h) {
    ContactEval ce;
    ce.hit = false;
    V3 rb = ld(m.sph_pos[s]);
    const float rad = m.sph_radius[s];
    float depth;
    V3 n;
    if (!columns_contact(ter, cs, rad, depth, n)) return ce;
    ce.nb = mulT(k.R, n);
    ce.rc = rb - rad * ce.nb;
    V3 vpb = k.v.l + cross(k.v.a, ce.rc);
    float vn = dot(vpb, ce.nb);
    float d_eff = depth < m.contact_max_pen ? depth : m.contact_max_pen;
    ce.cn = m.contact_cn;         // damper on approach AND on rebound (restitution 0); fn0 <= 0 below = no adhesion
    float fn0 = m.contact_kn * d_eff - ce.cn * vn;
    if (fn0 <= 0.f) return ce;
    V3 vt = vpb - vn * ce.nb;
    float vtn = p_sqrt(dot(vt, vt));
    ce.ct = m.contact_ct;
    float ct_cone = m.friction_mu * fn0 * p_rcp(vtn > 1e-4f ? vtn : 1e-4f);
    if (ct_cone < ce.ct) ce.ct = ct_cone;
    ce.F0 = fn0 * ce.nb - ce.ct * vt;
    // the known part of the contact point's world-velocity change, h w x v_centre (the body frame turns during the step): parc_sim_core.h pass1
    V3 wv = h * cross(k.v.a, k.v.l + cross(k.v.a, rb));
    float wvn = dot(wv, ce.nb);
    ce.F0 = ce.F0 - ((ce.cn + h * m.contact_kn) * wvn) * ce.nb - ce.ct * (wv - wvn * ce.nb);
    ce.hit = true;
    return ce;
}

#define BPL_CC_SLOTS 8      // contacts per body kept in LDS between the impedance pass and the force report
#define BPL_CC_FLOATS 11    // rc, nb, F0, cn, ct

__device__ __forceinline__ void substep(const parc_sim_model_t &m, const parc_terrain_t &ter, V3 env_off, const Lane &L, int b, int maxd,
                                        LState &x, float h, float cweight, float *lds /* [16][BPL_CONTRIB] of this env */,
                                        float *cc /* [BPL_CC_SLOTS][BPL_CC_FLOATS] of this lane */) {
    Kin k;
    kin_pass(L, b, maxd, x, k);
    const bool valid = L.depth >= 0;
    // ---- rigid-body inertia about the body origin, bias force, gravity
    SI IA;
    SV pA;
    {
        V3 hc = L.mass * L.com;
        IA.A.m[0] = L.io[0]; IA.A.m[1] = L.io[1]; IA.A.m[2] = L.io[2];
        IA.A.m[3] = L.io[1]; IA.A.m[4] = L.io[3]; IA.A.m[5] = L.io[4];
        IA.A.m[6] = L.io[2]; IA.A.m[7] = L.io[4]; IA.A.m[8] = L.io[5];
        IA.B = skew(hc);
        IA.C = ident(L.mass);
        SV Iv = mul(IA, k.v);
        pA.a = cross(k.v.a, Iv.a) + cross(k.v.l, Iv.l);
        pA.l = cross(k.v.a, Iv.l);
        V3 fg = mulT(k.R, v3(0.f, 0.f, -m.gravity * L.mass));
        pA.a = pA.a - cross(L.com, fg);
        pA.l = pA.l - fg;
        // per-link angular damping: the couple -c I_com w, with I_com w = I_o w + m c x (c x w)
        pA.a = pA.a + m.angular_damping * (mul(IA.A, k.v.a) + L.mass * cross(L.com, cross(L.com, k.v.a)));
    }
    int n_hit = 0;                    // active contacts of this body; the first BPL_CC_SLOTS are parked in LDS for the report
    unsigned long long overflow = 0ull;
    // ---- link-link contacts: every lane publishes its body's capsule and motion in the env frame (21 floats in the LDS exchange
    // buffer, free until the inward sweep), then tests it against the bodies of its self-collision set: the same pair is evaluated
    // by both lanes in canonical order (lower body index first), so the explicit forces are equal and opposite (link_contact)
    V3 flink = v3(0.f, 0.f, 0.f);
    {
        const CapsuleW mine = capsule_world(m, valid ? b : 0, k.R, k.P, k.v);
        float *o = lds + b * BPL_CONTRIB;
        o[0] = mine.a.x; o[1] = mine.a.y; o[2] = mine.a.z; o[3] = mine.b.x; o[4] = mine.b.y; o[5] = mine.b.z; o[6] = valid ? mine.r : 0.f;
        o[7] = mine.o.x; o[8] = mine.o.y; o[9] = mine.o.z; o[10] = mine.v.x; o[11] = mine.v.y; o[12] = mine.v.z;
        o[13] = mine.w.x; o[14] = mine.w.y; o[15] = mine.w.z; o[16] = L.mass;
        o[17] = mine.c.x; o[18] = mine.c.y; o[19] = mine.c.z; o[20] = mine.ext;
        __syncthreads();
        for (unsigned mm = (valid && mine.r > 0.f) ? m.self_mask[b] : 0u; mm; mm &= mm - 1) {
            const int ob = __ffs((int)mm) - 1;
            const float *q = lds + ob * BPL_CONTRIB;
            CapsuleW other;
            other.c = v3(q[17], q[18], q[19]); other.ext = q[20];
            {
                const V3 cc = mine.c - other.c;
                const float far = mine.ext + other.ext;
                if (!(q[6] > 0.f) || dot(cc, cc) >= far * far) continue;         // reject before touching the other 16 floats
            }
            other.a = v3(q[0], q[1], q[2]); other.b = v3(q[3], q[4], q[5]); other.r = q[6];
            other.o = v3(q[7], q[8], q[9]); other.v = v3(q[10], q[11], q[12]); other.w = v3(q[13], q[14], q[15]);
            LinkHit hit;
            if (!link_contact(m, mine, k.R, L.mass, other, q[16], h, b < ob, hit)) continue;
            pA.a = pA.a - hit.tau;
            pA.l = pA.l - hit.F;
            flink = flink + hit.F;
        }
        __syncthreads();                     // the exchange buffer is reused by the inward sweep
    }
    // ---- contacts of this body's sample spheres: implicit spring-damper + regularised friction
    // (software-pipelined by one: the heights under the NEXT sphere are loading while this one is evaluated - a foot lane walks 8 spheres
    // and each used to wait for its own loads)
    unsigned long long mm = valid ? L.sph : 0ull;
    ColumnSample cs_next;
    if (mm) cs_next = sample_sphere(m, ter, env_off, k, __ffsll((long long)mm) - 1);
    for (; mm; mm &= mm - 1) {
        const int s = __ffsll((long long)mm) - 1;
        const ColumnSample cs = cs_next;
        const unsigned long long rest = mm & (mm - 1);
        if (rest) cs_next = sample_sphere(m, ter, env_off, k, __ffsll((long long)rest) - 1);
        ContactEval ce = eval_contact(m, ter, cs, k, s, h);
        if (!ce.hit) continue;
        if (n_hit < BPL_CC_SLOTS) {
            float *o = cc + n_hit * BPL_CC_FLOATS;
            o[0] = ce.rc.x; o[1] = ce.rc.y; o[2] = ce.rc.z; o[3] = ce.nb.x; o[4] = ce.nb.y; o[5] = ce.nb.z;
            o[6] = ce.F0.x; o[7] = ce.F0.y; o[8] = ce.F0.z; o[9] = ce.cn; o[10] = ce.ct;
            ++n_hit;
        } else {
            overflow |= 1ull << s;
        }
        M3 Z = add(ident(ce.ct), outer(((ce.cn + h * m.contact_kn) - ce.ct) * ce.nb, ce.nb));
        M3 Sr = skew(ce.rc);
        M3 SZ = mul(Sr, Z);
        M3 SZS = mul(SZ, Sr);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            IA.A.m[q] -= h * SZS.m[q];
            IA.B.m[q] += h * SZ.m[q];
            IA.C.m[q] += h * Z.m[q];
        }
        pA.a = pA.a - cross(ce.rc, ce.F0);
        pA.l = pA.l - ce.F0;
    }
    // ---- inward sweep: joint drive (implicit PD + limits), articulated inertia reduction, hand-over to the parent
    M3 Ua, Ul, Dinv;
    V3 u = v3(0, 0, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q) Ua.m[q] = Ul.m[q] = Dinv.m[q] = 0.f;
    // drive torque and the diagonal augmentation of D depend on this joint's own state only: once per substep, not per level
    float tau[3] = {0.f, 0.f, 0.f}, aug[3] = {0.f, 0.f, 0.f};
    if (L.jt == PARC_JOINT_SPHERICAL) {
        V3 err = q_to_exp(qmul(qconj(x.jq), x.tq)) - h * x.jw;
        V3 e = q_to_exp(x.jq);
        const float ev[3] = {err.x, err.y, err.z}, wv[3] = {x.jw.x, x.jw.y, x.jw.z}, ee[3] = {e.x, e.y, e.z};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float kp = m.kp[L.d0 + q], kd = m.kd[L.d0 + q];
            float t = kp * ev[q] - kd * wv[q];
            float lim = m.effort[L.d0 + q];
            float sc = (lim > 0.f && fabsf(t) > lim) ? lim * p_rcp(fabsf(t)) : 1.0f;
            tau[q] = sc * t;
            aug[q] = m.armature[L.d0 + q] + sc * (h * kd + h * h * kp);
            float hi = m.limit_hi[L.d0 + q], lo = m.limit_lo[L.d0 + q];
            float over = ee[q] > hi ? ee[q] - hi : (ee[q] < lo ? ee[q] - lo : 0.f);
            if (over != 0.f) {
                tau[q] += -m.limit_kp * (over + h * wv[q]) - m.limit_kd * wv[q];
                aug[q] += h * m.limit_kd + h * h * m.limit_kp;
            }
        }
    } else if (L.jt == PARC_JOINT_HINGE) {
        float w = x.jw.x;
        float kp = m.kp[L.d0], kd = m.kd[L.d0];
        float t = kp * (x.tang - x.jang - h * w) - kd * w;
        float lim = m.effort[L.d0];
        float sc = (lim > 0.f && fabsf(t) > lim) ? lim * p_rcp(fabsf(t)) : 1.0f;
        tau[0] = sc * t;
        aug[0] = m.armature[L.d0] + sc * (h * kd + h * h * kp);
        float hi = m.limit_hi[L.d0], lo = m.limit_lo[L.d0];
        float over = x.jang > hi ? x.jang - hi : (x.jang < lo ? x.jang - lo : 0.f);
        if (over != 0.f) {
            tau[0] += -m.limit_kp * (over + h * w) - m.limit_kd * w;
            aug[0] += h * m.limit_kd + h * h * m.limit_kp;
        }
    }
    for (int l = maxd; l >= 1; --l) {
        if (L.depth == l) {
            SI Ia = IA;
            SV pa = pA;
            if (L.jt == PARC_JOINT_SPHERICAL) {
                M3 D = Ia.A;
                D.m[0] += aug[0]; D.m[4] += aug[1]; D.m[8] += aug[2];
                Dinv = inv_sym(D);
                Ua = Ia.A;
                Ul = transpose(Ia.B);
                u = v3(tau[0], tau[1], tau[2]) - pa.a;
                M3 UaDi = mul(Ua, Dinv), UlDi = mul(Ul, Dinv);
                Ia.A = sub(Ia.A, mulABt(UaDi, Ua));
                Ia.B = sub(Ia.B, mulABt(UaDi, Ul));
                Ia.C = sub(Ia.C, mulABt(UlDi, Ul));
                SV Iac = mul(Ia, k.c);
                V3 Diu = mul(Dinv, u);
                pa.a = pa.a + Iac.a + mul(Ua, Diu);
                pa.l = pa.l + Iac.l + mul(Ul, Diu);
            } else if (L.jt == PARC_JOINT_HINGE) {
                V3 ua = mul(Ia.A, L.ax), ul = mulT(Ia.B, L.ax);
                float D = dot(L.ax, ua) + aug[0];
                float Di = p_rcp(D);
                float uu = tau[0] - dot(L.ax, pa.a);
                Ua.m[0] = ua.x; Ua.m[1] = ua.y; Ua.m[2] = ua.z;
                Ul.m[0] = ul.x; Ul.m[1] = ul.y; Ul.m[2] = ul.z;
                Dinv.m[0] = Di;
                u = v3(uu, 0.f, 0.f);
                Ia.A = sub(Ia.A, outer(Di * ua, ua));
                Ia.B = sub(Ia.B, outer(Di * ua, ul));
                Ia.C = sub(Ia.C, outer(Di * ul, ul));
                SV Iac = mul(Ia, k.c);
                pa.a = pa.a + Iac.a + (Di * uu) * ua;
                pa.l = pa.l + Iac.l + (Di * uu) * ul;
            }
            // to the parent frame: rotate by E, shift by r
            M3 Ar = mulABt(mul(k.E, Ia.A), k.E), Br = mulABt(mul(k.E, Ia.B), k.E), Cr = mulABt(mul(k.E, Ia.C), k.E);
            M3 S = skew(L.r);
            M3 SC = mul(S, Cr);
            M3 SBt = mulABt(S, Br);
            M3 SCS = mul(SC, S);
            M3 Ap = sub(add(add(Ar, SBt), transpose(SBt)), SCS);
            M3 Bp = add(Br, SC);
            V3 fl = mul(k.E, pa.l);
            V3 fa = mul(k.E, pa.a) + cross(L.r, fl);
            float *o = lds + b * BPL_CONTRIB;
            o[0] = Ap.m[0]; o[1] = Ap.m[1]; o[2] = Ap.m[2]; o[3] = Ap.m[4]; o[4] = Ap.m[5]; o[5] = Ap.m[8];
#pragma unroll
            for (int q = 0; q < 9; ++q) o[6 + q] = Bp.m[q];
            o[15] = Cr.m[0]; o[16] = Cr.m[1]; o[17] = Cr.m[2]; o[18] = Cr.m[4]; o[19] = Cr.m[5]; o[20] = Cr.m[8];
            o[21] = fa.x; o[22] = fa.y; o[23] = fa.z; o[24] = fl.x; o[25] = fl.y; o[26] = fl.z;
        }
        __syncthreads();
        // parents one level up collect their children of level l
        const unsigned lvl = (unsigned)((__ballot(L.depth == l) >> (BPL_G * ((threadIdx.x & 63) / BPL_G))) & 0xFFFFull);
        for (unsigned cm = (valid ? L.children : 0u) & lvl; cm; cm &= cm - 1) {
            const float *o = lds + (__ffs((int)cm) - 1) * BPL_CONTRIB;
            IA.A.m[0] += o[0]; IA.A.m[1] += o[1]; IA.A.m[2] += o[2];
            IA.A.m[3] += o[1]; IA.A.m[4] += o[3]; IA.A.m[5] += o[4];
            IA.A.m[6] += o[2]; IA.A.m[7] += o[4]; IA.A.m[8] += o[5];
#pragma unroll
            for (int q = 0; q < 9; ++q) IA.B.m[q] += o[6 + q];
            IA.C.m[0] += o[15]; IA.C.m[1] += o[16]; IA.C.m[2] += o[17];
            IA.C.m[3] += o[16]; IA.C.m[4] += o[18]; IA.C.m[5] += o[19];
            IA.C.m[6] += o[17]; IA.C.m[7] += o[19]; IA.C.m[8] += o[20];
            pA.a = pA.a + v3(o[21], o[22], o[23]);
            pA.l = pA.l + v3(o[24], o[25], o[26]);
        }
        __syncthreads();
    }
    // ---- floating base (meaningful on lane 0): [[A,B],[B^T,C]] [alpha; a] = -[pn; pf]
    SV a;
    {
        M3 Ci = inv_sym(IA.C);
        M3 BCi = mul(IA.B, Ci);
        M3 Sc = sub(IA.A, mulABt(BCi, IA.B));
        M3 Sci = inv_sym(Sc);
        V3 alpha = mul(Sci, mul(BCi, pA.l) - pA.a);
        V3 lin = mul(Ci, -(pA.l + mulT(IA.B, alpha)));
        a = SV{alpha, lin};
    }
    // ---- outward sweep 2: accelerations, joint velocity update
    for (int l = 1; l <= maxd; ++l) {
        SV ap = shf(a, L.p);
        if (L.depth == l) {
            SV a1;
            a1.a = mulT(k.E, ap.a) + k.c.a;
            a1.l = mulT(k.E, ap.l + cross(ap.a, L.r)) + k.c.l;
            if (L.jt == PARC_JOINT_SPHERICAL) {
                V3 rhs = u - (mulT(Ua, a1.a) + mulT(Ul, a1.l));
                V3 qdd = mul(Dinv, rhs);
                a1.a = a1.a + qdd;
                x.jw = x.jw + h * qdd;
            } else if (L.jt == PARC_JOINT_HINGE) {
                V3 ua = v3(Ua.m[0], Ua.m[1], Ua.m[2]), ul = v3(Ul.m[0], Ul.m[1], Ul.m[2]);
                float qdd = Dinv.m[0] * (u.x - dot(ua, a1.a) - dot(ul, a1.l));
                a1.a = a1.a + qdd * L.ax;
                x.jw.x += h * qdd;
            }
            a = a1;
        }
    }
    // ---- realised contact force of this body, world frame (F+ = F0 - Z h J a), averaged over the substeps
    if (cweight > 0.f) {
        V3 Fb = v3(0, 0, 0);
        for (int c = 0; c < n_hit; ++c) {
            const float *o = cc + c * BPL_CC_FLOATS;
            const V3 rc = v3(o[0], o[1], o[2]), nb = v3(o[3], o[4], o[5]), F0 = v3(o[6], o[7], o[8]);
            V3 dv = h * (a.l + cross(a.a, rc));
            float dvn = dot(dv, nb);
            V3 F = F0 - (o[9] + h * m.contact_kn) * dvn * nb - o[10] * (dv - dvn * nb);
            float fnn = dot(F, nb);
            if (fnn < 0.f) F = F - fnn * nb;
            Fb = Fb + F;
        }
        for (unsigned long long mm = overflow; mm; mm &= mm - 1) {      // more simultaneous contacts than slots: re-evaluate
            const int so = __ffsll((long long)mm) - 1;
            ContactEval ce = eval_contact(m, ter, sample_sphere(m, ter, env_off, k, so), k, so, h);
            V3 dv = h * (a.l + cross(a.a, ce.rc));
            float dvn = dot(dv, ce.nb);
            V3 F = ce.F0 - (ce.cn + h * m.contact_kn) * dvn * ce.nb - ce.ct * (dv - dvn * ce.nb);
            float fnn = dot(F, ce.nb);
            if (fnn < 0.f) F = F - fnn * ce.nb;
            Fb = Fb + F;
        }
        x.cforce = x.cforce + cweight * mul(k.R, Fb + flink);
    }
    // ---- semi-implicit Euler
    const float wmax = m.max_angular_velocity;
    if (b == 0) {
        // linear velocity in WORLD coordinates, v_w += h R (a_sp + w x v): parc_sim_core.h, substep
        const V3 vw_new = mul(k.R, x.root_vel.l + h * (a.l + cross(x.root_vel.a, x.root_vel.l)));
        x.root_vel.a = x.root_vel.a + h * a.a;
        float wn = p_sqrt(dot(x.root_vel.a, x.root_vel.a));
        if (wn > wmax) x.root_vel.a = (wmax * p_rcp(wn)) * x.root_vel.a;
        x.root_rot = qnormalize(qmul(x.root_rot, exp_to_q(h * x.root_vel.a)));
        x.root_pos = x.root_pos + h * vw_new;
        x.root_vel.l = mulT(qmat(x.root_rot), vw_new);
    } else if (L.jt == PARC_JOINT_SPHERICAL) {
        float wn = p_sqrt(dot(x.jw, x.jw));
        if (wn > wmax) x.jw = (wmax * p_rcp(wn)) * x.jw;
        x.jq = qnormalize(qmul(x.jq, exp_to_q(h * x.jw)));
    } else if (L.jt == PARC_JOINT_HINGE) {
        x.jw.x = clampf(x.jw.x, -wmax, wmax);
        x.jang += h * x.jw.x;
        x.jq = exp_to_q(x.jang * L.ax);
    }
}

__device__ __forceinline__ void load_lane_state(const parc_sim_model_t &m, const Lane &L, int b, const float *root_state, const float *dof_state,
                                                const float *action, const float *act_lo, const float *act_hi, LState &x) {
    x.root_pos = ld(root_state);
    x.root_rot = qnormalize(Q4{root_state[3], root_state[4], root_state[5], root_state[6]});
    M3 R = qmat(x.root_rot);
    x.root_vel.l = mulT(R, ld(root_state + 7));
    x.root_vel.a = mulT(R, ld(root_state + 10));
    x.jq = Q4{0.f, 0.f, 0.f, 1.f};
    x.tq = Q4{0.f, 0.f, 0.f, 1.f};
    x.jang = 0.f;
    x.tang = 0.f;
    x.jw = v3(0, 0, 0);
    x.cforce = v3(0, 0, 0);
    const int d0 = L.d0;
    if (L.jt == PARC_JOINT_SPHERICAL) {
        x.jq = exp_to_q(v3(dof_state[2 * d0], dof_state[2 * (d0 + 1)], dof_state[2 * (d0 + 2)]));
        x.jw = v3(dof_state[2 * d0 + 1], dof_state[2 * (d0 + 1) + 1], dof_state[2 * (d0 + 2) + 1]);
        float t[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) t[q] = clampf(action[d0 + q], act_lo[d0 + q], act_hi[d0 + q]);   // ig_char_env.py:490
        x.tq = exp_to_q(v3(t[0], t[1], t[2]));
    } else if (L.jt == PARC_JOINT_HINGE) {
        x.jang = dof_state[2 * d0];
        x.jw.x = dof_state[2 * d0 + 1];
        x.jq = exp_to_q(x.jang * L.ax);
        x.tang = clampf(action[d0], act_lo[d0], act_hi[d0]);
    }
}

// state rows + body poses / velocities of the final state (one more outward sweep, carrying the world quaternion)
template <bool WRITE_STATE = true>
__device__ __forceinline__ void store_lane_state(const Lane &L, int b, int maxd, const LState &x, float *root_state, float *dof_state,
                                                 float *rigid_body_state, float *contact_forces) {
    Kin k;
    kin_pass(L, b, maxd, x, k);
    // world quaternion: Q_b = normalize(Q_parent * (local_rotation * jq)), the chain of the reference's publish_bodies
    const Q4 lq = qmul(L.lrot, x.jq);
    Q4 Q = x.root_rot;
    for (int l = 1; l <= maxd; ++l) {
        Q4 Qp = Q4{shf(Q.x, L.p), shf(Q.y, L.p), shf(Q.z, L.p), shf(Q.w, L.p)};
        if (L.depth == l) Q = qnormalize(qmul(Qp, lq));
    }
    if (!WRITE_STATE) {
        // refresh after a reset: only the body poses / velocities are published, the state rows stay as written
    } else if (b == 0) {
        M3 R = qmat(x.root_rot);
        st(root_state, x.root_pos);
        root_state[3] = x.root_rot.x; root_state[4] = x.root_rot.y; root_state[5] = x.root_rot.z; root_state[6] = x.root_rot.w;
        st(root_state + 7, mul(R, x.root_vel.l));
        st(root_state + 10, mul(R, x.root_vel.a));
    } else if (L.jt == PARC_JOINT_SPHERICAL) {
        V3 e = q_to_exp(x.jq);
        const int d0 = L.d0;
        dof_state[2 * d0] = e.x; dof_state[2 * (d0 + 1)] = e.y; dof_state[2 * (d0 + 2)] = e.z;
        dof_state[2 * d0 + 1] = x.jw.x; dof_state[2 * (d0 + 1) + 1] = x.jw.y; dof_state[2 * (d0 + 2) + 1] = x.jw.z;
    } else if (L.jt == PARC_JOINT_HINGE) {
        dof_state[2 * L.d0] = x.jang;
        dof_state[2 * L.d0 + 1] = x.jw.x;
    }
    if (L.depth >= 0) {
        float *o = rigid_body_state + 13 * b;
        st(o, k.P);
        o[3] = Q.x; o[4] = Q.y; o[5] = Q.z; o[6] = Q.w;
        st(o + 7, mul(k.R, k.v.l));
        st(o + 10, mul(k.R, k.v.a));
        st(contact_forces + 3 * b, x.cforce);
    }
}

// The whole env step as seen by lane b of an env's 16-lane group: what sim_step_bpl_kernel runs (and what the host lane
// emulation runs, lane by lane in lock step).  `lds`: BPL_G * BPL_CONTRIB floats shared by the group, `cc`: this lane's contact cache.
__device__ __forceinline__ void step_lane(const parc_sim_model_t &m, const parc_terrain_t &ter, int b, float *root_state, float *dof_state,
                                          float *rigid_body_state, float *contact_forces, const float *env_offset, const float *action,
                                          const float *act_lo, const float *act_hi, int n_sub, float h, float *lds, float *cc) {
    const Lane L = load_lane(m, b);
    int maxd = L.depth;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        const int other = __shfl_xor(maxd, o, BPL_G);
        maxd = other > maxd ? other : maxd;
    }
    LState x;
    load_lane_state(m, L, b, root_state, dof_state, action, act_lo, act_hi, x);
    const V3 off = ld(env_offset);
    const float w = 1.0f / (float)n_sub;
    for (int s = 0; s < n_sub; ++s) substep(m, ter, off, L, b, maxd, x, h, w, lds, cc);
    store_lane_state(L, b, maxd, x, root_state, dof_state, rigid_body_state, contact_forces);
}

}  // namespace parc_sim_bpl
