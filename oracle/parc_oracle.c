/*
 * parc_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp32 like the reference) of the kinematic / observation / reward /
 * termination / TD(lambda) half of the PARC tracker hot path.  Every function cites the reference
 * file:line it follows (paths relative to the reference root).  It is pinned by the .npz fixtures under tests/golden,
 * which were produced by the reference's own Python on CPU (tests/golden/gen_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (parc_amd/) never links, imports or calls it.
 *
 * The dynamics half (Isaac Gym / PhysX) has no arithmetic reference: parity unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define JT_ROOT 0
#define JT_HINGE 1
#define JT_SPHERICAL 2
#define JT_FIXED 3

/* ------------------------------------------------------------------ quaternion helpers (xyzw) */

/* util/torch_util.py:9-12 normalize(x, eps=1e-9) */
static void vnormalize(const float *x, int n, float *out) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += x[i] * x[i];
    float nrm = sqrtf(s);
    if (nrm < 1e-9f) nrm = 1e-9f;
    for (int i = 0; i < n; ++i) out[i] = x[i] / nrm;
}

/* util/torch_util.py:4-7 */
static float normalize_angle(float x) { return atan2f(sinf(x), cosf(x)); }

/* util/torch_util.py:40-58 (same operation order) */
static void quat_mul(const float *a, const float *b, float *o) {
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float w = qq - ww + (z1 - y1) * (y2 - z2);
    float x = qq - xx + (x1 + w1) * (x2 + w2);
    float y = qq - yy + (w1 - x1) * (y2 + z2);
    float z = qq - zz + (z1 + y1) * (w2 - x2);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

static void cross3(const float *a, const float *b, float *o) {
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

/* util/torch_util.py:60-66 */
static void quat_rotate(const float *q, const float *v, float *o) {
    float t[3], c[3];
    cross3(q, v, t);
    t[0] *= 2.f; t[1] *= 2.f; t[2] *= 2.f;
    cross3(q, t, c);
    float r0 = v[0] + q[3] * t[0] + c[0];
    float r1 = v[1] + q[3] * t[1] + c[1];
    float r2 = v[2] + q[3] * t[2] + c[2];
    o[0] = r0; o[1] = r1; o[2] = r2;
}

/* util/torch_util.py:29-31 */
static void quat_conj(const float *q, float *o) { o[0] = -q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = q[3]; }

/* util/torch_util.py:33-38 */
static void quat_pos(const float *q, float *o) {
    float s = (q[3] < 0.f) ? -1.f : 1.f;
    for (int i = 0; i < 4; ++i) o[i] = s * q[i];
}

/* util/torch_util.py:311-317 */
static void axis_angle_to_quat(const float *axis, float angle, float *o) {
    float theta = angle / 2.f;
    float ax[3];
    vnormalize(axis, 3, ax);
    float s = sinf(theta);
    float q[4] = {ax[0] * s, ax[1] * s, ax[2] * s, cosf(theta)};
    vnormalize(q, 4, o);
}

/* util/torch_util.py:394-412 */
static void exp_map_to_axis_angle(const float *em, float *axis, float *angle) {
    float a = sqrtf(em[0] * em[0] + em[1] * em[1] + em[2] * em[2]);
    float ax[3] = {em[0] / a, em[1] / a, em[2] / a};
    a = normalize_angle(a);
    if (fabsf(a) > 1e-5f) {
        axis[0] = ax[0]; axis[1] = ax[1]; axis[2] = ax[2];
        *angle = a;
    } else {
        axis[0] = 0.f; axis[1] = 0.f; axis[2] = 1.f;
        *angle = 0.f;
    }
}

/* util/torch_util.py:414-419 */
static void exp_map_to_quat(const float *em, float *o) {
    float axis[3], angle;
    exp_map_to_axis_angle(em, axis, &angle);
    axis_angle_to_quat(axis, angle, o);
}

/* util/torch_util.py:68-88 */
static void quat_to_axis_angle(const float *q_in, float *axis, float *angle) {
    float q[4];
    quat_pos(q_in, q);
    float length = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float a = 2.0f * atan2f(length, q[3]);
    if (length > 1e-5f) {
        axis[0] = q[0] / length; axis[1] = q[1] / length; axis[2] = q[2] / length;
        *angle = a;
    } else {
        axis[0] = 0.f; axis[1] = 0.f; axis[2] = 1.f;
        *angle = 0.f;
    }
}

/* util/torch_util.py:346-351 */
static void quat_to_exp_map(const float *q, float *o) {
    float axis[3], angle;
    quat_to_axis_angle(q, axis, &angle);
    o[0] = angle * axis[0]; o[1] = angle * axis[1]; o[2] = angle * axis[2];
}

/* util/torch_util.py:427-431 */
static float quat_diff_angle(const float *q0, const float *q1) {
    float c[4], dq[4], axis[3], angle;
    quat_conj(q0, c);
    quat_mul(q1, c, dq);
    quat_to_axis_angle(dq, axis, &angle);
    return angle;
}

/* util/torch_util.py:361-373 */
static void quat_to_tan_norm(const float *q, float *o) {
    const float ex[3] = {1.f, 0.f, 0.f}, ez[3] = {0.f, 0.f, 1.f};
    quat_rotate(q, ex, o);
    quat_rotate(q, ez, o + 3);
}

/* util/torch_util.py:443-468 */
static void slerp(const float *q0, const float *q1_in, float t, float *o) {
    float c = q0[0] * q1_in[0] + q0[1] * q1_in[1] + q0[2] * q1_in[2] + q0[3] * q1_in[3];
    float q1[4];
    float sgn = (c < 0.f) ? -1.f : 1.f;
    for (int i = 0; i < 4; ++i) q1[i] = sgn * q1_in[i];
    c = fabsf(c);
    float half_theta = acosf(c);
    float s = sqrtf(1.0f - c * c);
    float ra = sinf((1.f - t) * half_theta) / s;
    float rb = sinf(t * half_theta) / s;
    for (int i = 0; i < 4; ++i) {
        float v = ra * q0[i] + rb * q1[i];
        if (fabsf(s) < 0.001f) v = 0.5f * q0[i] + 0.5f * q1[i];
        if (fabsf(c) >= 1.f) v = q0[i];
        o[i] = v;
    }
}

/* util/torch_util.py:470-479 */
static float calc_heading(const float *q) {
    const float ex[3] = {1.f, 0.f, 0.f};
    float d[3];
    quat_rotate(q, ex, d);
    return atan2f(d[1], d[0]);
}

/* util/torch_util.py:491-499 */
static void calc_heading_quat_inv(const float *q, float *o) {
    const float ez[3] = {0.f, 0.f, 1.f};
    axis_angle_to_quat(ez, -calc_heading(q), o);
}

/* ------------------------------------------------------------------ exported batch wrappers (G1) */
void orc_quat_mul(int n, const float *a, const float *b, float *o) { for (int i = 0; i < n; ++i) quat_mul(a + 4 * i, b + 4 * i, o + 4 * i); }
void orc_quat_rotate(int n, const float *q, const float *v, float *o) { for (int i = 0; i < n; ++i) quat_rotate(q + 4 * i, v + 3 * i, o + 3 * i); }
void orc_exp_map_to_quat(int n, const float *e, float *o) { for (int i = 0; i < n; ++i) exp_map_to_quat(e + 3 * i, o + 4 * i); }
void orc_quat_to_exp_map(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) quat_to_exp_map(q + 4 * i, o + 3 * i); }
void orc_axis_angle_to_quat(int n, const float *ax, const float *an, float *o) { for (int i = 0; i < n; ++i) axis_angle_to_quat(ax + 3 * i, an[i], o + 4 * i); }
void orc_quat_to_tan_norm(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) quat_to_tan_norm(q + 4 * i, o + 6 * i); }
void orc_slerp(int n, const float *a, const float *b, const float *t, float *o) { for (int i = 0; i < n; ++i) slerp(a + 4 * i, b + 4 * i, t[i], o + 4 * i); }
void orc_calc_heading(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) o[i] = calc_heading(q + 4 * i); }
void orc_calc_heading_quat_inv(int n, const float *q, float *o) { for (int i = 0; i < n; ++i) calc_heading_quat_inv(q + 4 * i, o + 4 * i); }
void orc_quat_diff_angle(int n, const float *a, const float *b, float *o) { for (int i = 0; i < n; ++i) o[i] = quat_diff_angle(a + 4 * i, b + 4 * i); }

/* ------------------------------------------------------------------ character model (a3-a5) */
typedef struct {
    int nb;                 /* bodies (15) */
    const int *parent;      /* [nb] */
    const float *ltrans;    /* [nb,3]  anim/kin_char_model.py:160-164 */
    const float *lrot;      /* [nb,4] */
    const int *jtype;       /* [nb]    JointType */
    const float *jaxis;     /* [nb,3]  hinge axis */
    const int *dof_idx;     /* [nb] */
} orc_char_t;

static int dof_dim(int jt) { return jt == JT_HINGE ? 1 : (jt == JT_SPHERICAL ? 3 : 0); }

/* anim/kin_char_model.py:57-77 + :478-491; one pose: dof[D] -> joint_rot[nb-1,4] */
static void dof_to_rot1(const orc_char_t *c, const float *dof, float *jrot) {
    for (int j = 1; j < c->nb; ++j) {
        float *o = jrot + 4 * (j - 1);
        int jt = c->jtype[j];
        if (jt == JT_HINGE) {
            axis_angle_to_quat(c->jaxis + 3 * j, dof[c->dof_idx[j]], o);
        } else if (jt == JT_SPHERICAL) {
            exp_map_to_quat(dof + c->dof_idx[j], o);
        } else {
            o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 1.f;
        }
    }
}

/* anim/kin_char_model.py:79-100 + :493-507 */
static void rot_to_dof1(const orc_char_t *c, const float *jrot, float *dof, int dof_size) {
    for (int d = 0; d < dof_size; ++d) dof[d] = 0.f;
    for (int j = 1; j < c->nb; ++j) {
        const float *q = jrot + 4 * (j - 1);
        int jt = c->jtype[j];
        if (jt == JT_HINGE) {
            float axis[3], angle;
            quat_to_axis_angle(q, axis, &angle);
            const float *ja = c->jaxis + 3 * j;
            float dot = ja[0] * axis[0] + ja[1] * axis[1] + ja[2] * axis[2];
            if (dot < 0.f) angle *= -1.f;
            dof[c->dof_idx[j]] = angle;
        } else if (jt == JT_SPHERICAL) {
            quat_to_exp_map(q, dof + c->dof_idx[j]);
        }
    }
}

/* anim/kin_char_model.py:509-541 */
static void forward_kinematics1(const orc_char_t *c, const float *root_pos, const float *root_rot,
                                const float *jrot, float *body_pos, float *body_rot) {
    for (int k = 0; k < 3; ++k) body_pos[k] = root_pos[k];
    for (int k = 0; k < 4; ++k) body_rot[k] = root_rot[k];
    for (int j = 1; j < c->nb; ++j) {
        int p = c->parent[j];
        float wt[3], cr[4];
        quat_rotate(body_rot + 4 * p, c->ltrans + 3 * j, wt);
        for (int k = 0; k < 3; ++k) body_pos[3 * j + k] = body_pos[3 * p + k] + wt[k];
        quat_mul(c->lrot + 4 * j, jrot + 4 * (j - 1), cr);
        quat_mul(body_rot + 4 * p, cr, body_rot + 4 * j);
    }
}

static orc_char_t mk_char(int nb, const int *parent, const float *ltrans, const float *lrot, const int *jtype,
                          const float *jaxis, const int *dof_idx) {
    orc_char_t c = {nb, parent, ltrans, lrot, jtype, jaxis, dof_idx};
    return c;
}

static int char_dof_size(const orc_char_t *c) {
    int d = 0;
    for (int j = 0; j < c->nb; ++j) d += dof_dim(c->jtype[j]);
    return d;
}

#define CHAR_ARGS int nb, const int *parent, const float *ltrans, const float *lrot, const int *jtype, const float *jaxis, const int *dof_idx
#define CHAR_PASS nb, parent, ltrans, lrot, jtype, jaxis, dof_idx

void orc_dof_to_rot(CHAR_ARGS, int n, const float *dof, float *jrot) {
    orc_char_t c = mk_char(CHAR_PASS);
    int D = char_dof_size(&c);
    for (int i = 0; i < n; ++i) dof_to_rot1(&c, dof + (size_t)i * D, jrot + (size_t)i * (nb - 1) * 4);
}

void orc_rot_to_dof(CHAR_ARGS, int n, const float *jrot, float *dof) {
    orc_char_t c = mk_char(CHAR_PASS);
    int D = char_dof_size(&c);
    for (int i = 0; i < n; ++i) rot_to_dof1(&c, jrot + (size_t)i * (nb - 1) * 4, dof + (size_t)i * D, D);
}

void orc_forward_kinematics(CHAR_ARGS, int n, const float *root_pos, const float *root_rot, const float *jrot,
                            float *body_pos, float *body_rot) {
    orc_char_t c = mk_char(CHAR_PASS);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        forward_kinematics1(&c, root_pos + 3 * (size_t)i, root_rot + 4 * (size_t)i, jrot + (size_t)i * (nb - 1) * 4,
                            body_pos + (size_t)i * nb * 3, body_rot + (size_t)i * nb * 4);
}

/* ------------------------------------------------------------------ motion library (a6) */

/*
 * anim/motion_lib.py:264-290,405-423 and anim/kin_char_model.py:543-581 -- derive the per-frame
 * arrays of ONE clip from its [F,6+D] frames.  fps as in the file; dt = 1/fps.
 */
void orc_motion_derive(CHAR_ARGS, int F, const float *frames, double fps,
                       float *root_pos, float *root_rot, float *joint_rot,
                       float *root_vel, float *root_ang_vel, float *dof_vel) {
    orc_char_t c = mk_char(CHAR_PASS);
    int D = char_dof_size(&c);
    int J = nb - 1;
    int W = 6 + D;
    float fpsf = (float)fps;
    float dtf = (float)(1.0 / fps);
    for (int f = 0; f < F; ++f) {
        const float *fr = frames + (size_t)f * W;
        for (int k = 0; k < 3; ++k) root_pos[3 * f + k] = fr[k];
        exp_map_to_quat(fr + 3, root_rot + 4 * f);
        float *jr = joint_rot + (size_t)f * J * 4;
        dof_to_rot1(&c, fr + 6, jr);
        for (int j = 0; j < J; ++j) quat_pos(jr + 4 * j, jr + 4 * j); /* motion_lib.py:421 */
    }
    for (int f = 0; f + 1 < F; ++f) {
        for (int k = 0; k < 3; ++k) root_vel[3 * f + k] = fpsf * (root_pos[3 * (f + 1) + k] - root_pos[3 * f + k]);
        /* quat_diff(q0,q1) = q1 * conj(q0)  (torch_util.py:422-425) */
        float cj[4], dq[4], em[3];
        quat_conj(root_rot + 4 * f, cj);
        quat_mul(root_rot + 4 * (f + 1), cj, dq);
        quat_to_exp_map(dq, em);
        for (int k = 0; k < 3; ++k) root_ang_vel[3 * f + k] = fpsf * em[k];
        /* kin_char_model.py:552-581: drot = conj(q0) * q1, quat_normalize, exp map / dt */
        float *dv = dof_vel + (size_t)f * D;
        for (int d = 0; d < D; ++d) dv[d] = 0.f;
        for (int j = 1; j < nb; ++j) {
            const float *q0 = joint_rot + ((size_t)f * J + (j - 1)) * 4;
            const float *q1 = joint_rot + ((size_t)(f + 1) * J + (j - 1)) * 4;
            float c0[4], dr[4], dp[4], dn[4], e[3];
            quat_conj(q0, c0);
            quat_mul(c0, q1, dr);
            quat_pos(dr, dp);
            vnormalize(dp, 4, dn);
            int jt = jtype[j];
            if (jt == JT_HINGE) {
                quat_to_exp_map(dn, e);
                const float *ja = jaxis + 3 * j;
                dv[dof_idx[j]] = ja[0] * (e[0] / dtf) + ja[1] * (e[1] / dtf) + ja[2] * (e[2] / dtf);
            } else if (jt == JT_SPHERICAL) {
                quat_to_exp_map(dn, e);
                for (int k = 0; k < 3; ++k) dv[dof_idx[j] + k] = e[k] / dtf;
            }
        }
    }
    if (F >= 2) {
        for (int k = 0; k < 3; ++k) {
            root_vel[3 * (F - 1) + k] = root_vel[3 * (F - 2) + k];
            root_ang_vel[3 * (F - 1) + k] = root_ang_vel[3 * (F - 2) + k];
        }
        memcpy(dof_vel + (size_t)(F - 1) * D, dof_vel + (size_t)(F - 2) * D, sizeof(float) * D);
    }
}

typedef struct {
    int M, J, D, B;
    const int64_t *num_frames;  /* [M] */
    const int64_t *start_idx;   /* [M] */
    const float *length;        /* [M] */
    const int *loop_mode;       /* [M] 0 clamp 1 wrap */
    const float *pos_delta;     /* [M,3] */
    const float *root_pos, *root_rot, *joint_rot, *root_vel, *root_ang_vel, *dof_vel, *contacts;
} orc_mlib_t;

/* anim/motion_lib.py:80-112,443-475,527-538 -- one query */
static void calc_motion_frame1(const orc_mlib_t *m, int64_t id, float time, float *root_pos, float *root_rot,
                               float *root_vel, float *root_ang_vel, float *joint_rot, float *dof_vel,
                               float *contacts) {
    float len = m->length[id];
    int wrap = (m->loop_mode[id] == 1);
    float phase = time / len;
    if (wrap) phase = phase - floorf(phase);
    if (phase < 0.f) phase = 0.f;
    if (phase > 1.f) phase = 1.f;
    int64_t nf = m->num_frames[id];
    float fp = phase * (float)(nf - 1);
    int64_t i0 = (int64_t)fp; /* .long(): truncation */
    int64_t i1 = i0 + 1 < nf - 1 ? i0 + 1 : nf - 1;
    float blend = fp - (float)i0;
    i0 += m->start_idx[id];
    i1 += m->start_idx[id];
    for (int k = 0; k < 3; ++k)
        root_pos[k] = (1.0f - blend) * m->root_pos[3 * i0 + k] + blend * m->root_pos[3 * i1 + k];
    slerp(m->root_rot + 4 * i0, m->root_rot + 4 * i1, blend, root_rot);
    for (int k = 0; k < 3; ++k) {
        root_vel[k] = m->root_vel[3 * i0 + k];
        root_ang_vel[k] = m->root_ang_vel[3 * i0 + k];
    }
    for (int j = 0; j < m->J; ++j)
        slerp(m->joint_rot + ((size_t)i0 * m->J + j) * 4, m->joint_rot + ((size_t)i1 * m->J + j) * 4, blend, joint_rot + 4 * j);
    for (int d = 0; d < m->D; ++d) dof_vel[d] = m->dof_vel[(size_t)i0 * m->D + d];
    if (wrap) { /* _calc_loop_offset :458-475 */
        float ph = floorf(time / len);
        for (int k = 0; k < 3; ++k) root_pos[k] += ph * m->pos_delta[3 * id + k];
    } else {
        for (int k = 0; k < 3; ++k) root_pos[k] += 0.f;
    }
    if (contacts)
        for (int b = 0; b < m->B; ++b)
            contacts[b] = (1.0f - blend) * m->contacts[(size_t)i0 * m->B + b] + blend * m->contacts[(size_t)i1 * m->B + b];
}

#define MLIB_ARGS int M, int J, int D, int B, const int64_t *num_frames, const int64_t *start_idx, const float *length, \
    const int *loop_mode, const float *pos_delta, const float *f_root_pos, const float *f_root_rot,                  \
    const float *f_joint_rot, const float *f_root_vel, const float *f_root_ang_vel, const float *f_dof_vel,          \
    const float *f_contacts
#define MLIB_PASS M, J, D, B, num_frames, start_idx, length, loop_mode, pos_delta, f_root_pos, f_root_rot, f_joint_rot, f_root_vel, \
    f_root_ang_vel, f_dof_vel, f_contacts
#define MLIB_MAKE                                                                                                      \
    orc_mlib_t ml = {M, J, D, B, num_frames, start_idx, length, loop_mode, pos_delta, f_root_pos, f_root_rot,          \
                     f_joint_rot, f_root_vel, f_root_ang_vel, f_dof_vel, f_contacts}

void orc_calc_motion_frame(MLIB_ARGS, int Q, const int64_t *ids, const float *times, float *root_pos, float *root_rot,
                           float *root_vel, float *root_ang_vel, float *joint_rot, float *dof_vel, float *contacts) {
    MLIB_MAKE;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q)
        calc_motion_frame1(&ml, ids[q], times[q], root_pos + 3 * (size_t)q, root_rot + 4 * (size_t)q,
                           root_vel + 3 * (size_t)q, root_ang_vel + 3 * (size_t)q, joint_rot + (size_t)q * J * 4,
                           dof_vel + (size_t)q * D, contacts + (size_t)q * B);
}

/* Diagnostic for the parity tests (not a reference function): |cos(half angle)| between the two frames a query blends, as slerp
 * (util/torch_util.py:447-451) evaluates it in fp32, for the root rotation (column 0) and every joint (columns 1..J), plus the blend.
 * slerp has two value discontinuities in that cosine: `cos >= 1 -> q0` (:466) and `sin < 0.001 -> plain average` (:465); a query whose
 * cosine sits within an ulp or two of either is where a 1-ulp difference in a stored frame legitimately flips the branch. */
void orc_slerp_cosines(MLIB_ARGS, int Q, const int64_t *ids, const float *times, float *out_cos, float *out_blend) {
    MLIB_MAKE;
#pragma omp parallel for schedule(static)
    for (int q = 0; q < Q; ++q) {
        const orc_mlib_t *m = &ml;
        int64_t id = ids[q];
        float len = m->length[id];
        int wrap = (m->loop_mode[id] == 1);
        float phase = times[q] / len;
        if (wrap) phase = phase - floorf(phase);
        if (phase < 0.f) phase = 0.f;
        if (phase > 1.f) phase = 1.f;
        int64_t nf = m->num_frames[id];
        float fp = phase * (float)(nf - 1);
        int64_t i0 = (int64_t)fp;
        int64_t i1 = i0 + 1 < nf - 1 ? i0 + 1 : nf - 1;
        out_blend[q] = fp - (float)i0;
        i0 += m->start_idx[id];
        i1 += m->start_idx[id];
        for (int j = 0; j <= m->J; ++j) {
            const float *a = j == 0 ? m->root_rot + 4 * i0 : m->joint_rot + ((size_t)i0 * m->J + (j - 1)) * 4;
            const float *b = j == 0 ? m->root_rot + 4 * i1 : m->joint_rot + ((size_t)i1 * m->J + (j - 1)) * 4;
            float c = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
            out_cos[(size_t)q * (m->J + 1) + j] = fabsf(c);
        }
    }
}

/* ------------------------------------------------------------------ heightfield (a8) */

/* util/geom_util.py:249-270 + torch_util.rotate_2d_vec :619-631 ; out [num_rays*(num_neg+num_pos+1), 2] */
void orc_xy_points_cone(float dx, int num_neg, int num_pos, int rays_neg, int rays_pos, float angle_between, float *out) {
    int dim = num_neg + num_pos + 1;
    int num_rays = rays_neg + 1 + rays_pos;
    float start = -dx * (float)num_neg, end = dx * (float)num_pos;
    /* torch.linspace(start,end,dim) in fp32: step = (end-start)/(dim-1); symmetric evaluation from both ends */
    float step = (end - start) / (float)(dim - 1);
    int half = dim / 2;
    for (int r = 0; r < num_rays; ++r) {
        float ang = -angle_between * (float)(rays_neg - r);
        float c = cosf(ang), s = sinf(ang);
        for (int i = 0; i < dim; ++i) {
            float x = (i < half) ? start + step * (float)i : end - step * (float)(dim - i - 1);
            float y = 0.f;
            out[2 * (r * dim + i) + 0] = x * c - y * s;
            out[2 * (r * dim + i) + 1] = x * s + y * c;
        }
    }
}

/* util/terrain_util.py:107-126 round-half-even, clamp; :1329-1346 */
static float hf_lookup(const float *hf, int dim_x, int dim_y, float min_x, float min_y, float dx, float dy, float px, float py) {
    float fi = rintf((px - min_x) / dx);
    float fj = rintf((py - min_y) / dy);
    int64_t i = (int64_t)fi, j = (int64_t)fj;
    if (i < 0) i = 0;
    if (i > dim_x - 1) i = dim_x - 1;
    if (j < 0) j = 0;
    if (j > dim_y - 1) j = dim_y - 1;
    return hf[i * dim_y + j];
}

/* envs/ig_parkour/mgdm_dm_util.py:158-179 ; root_pos is the GLOBAL xyz (env offset already added,
 * ig_parkour_env.py:640), heading = calc_heading(root_rot).  out [N,P] */
void orc_refresh_ray_obs_hfs(int N, int P, const float *ray_xy, const float *root_pos, const float *heading,
                             const float *hf, int dim_x, int dim_y, float min_x, float min_y, float dx, float dy,
                             float min_h, float max_h, float *out) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        float c = cosf(heading[e]), s = sinf(heading[e]);
        float rx = root_pos[3 * e + 0], ry = root_pos[3 * e + 1], rz = root_pos[3 * e + 2];
        for (int p = 0; p < P; ++p) {
            float x = ray_xy[2 * p], y = ray_xy[2 * p + 1];
            float px = (x * c - y * s) + rx;
            float py = (x * s + y * c) + ry;
            float h = hf_lookup(hf, dim_x, dim_y, min_x, min_y, dx, dy, px, py) - rz;
            if (h < min_h) h = min_h;
            if (h > max_h) h = max_h;
            out[(size_t)e * P + p] = h;
        }
    }
}

/* ------------------------------------------------------------------ observations (a9) */

/* envs/ig_char_env.py:582-626 without the root-height column (the caller prepends it, :621-623); out[6+3+3+6J+D+3K].
 * global_obs (:587-590,:605): root rotation / velocities and the key-body offsets stay in world axes. */
static void char_obs1(int J, int D, int K, int global_obs, const float *root_pos, const float *root_rot, const float *root_vel,
                      const float *root_ang_vel, const float *joint_rot, const float *dof_vel, const float *key_pos,
                      float *o) {
    float h[4], lr[4];
    calc_heading_quat_inv(root_rot, h);
    if (global_obs) {
        quat_to_tan_norm(root_rot, o);
        for (int a = 0; a < 3; ++a) { o[6 + a] = root_vel[a]; o[9 + a] = root_ang_vel[a]; }
    } else {
        quat_mul(h, root_rot, lr);
        quat_to_tan_norm(lr, o);
        quat_rotate(h, root_vel, o + 6);
        quat_rotate(h, root_ang_vel, o + 9);
    }
    for (int j = 0; j < J; ++j) quat_to_tan_norm(joint_rot + 4 * j, o + 12 + 6 * j);
    for (int d = 0; d < D; ++d) o[12 + 6 * J + d] = dof_vel[d];
    for (int k = 0; k < K; ++k) {
        float rel[3] = {key_pos[3 * k] - root_pos[0], key_pos[3 * k + 1] - root_pos[1], key_pos[3 * k + 2] - root_pos[2]};
        float *ok = o + 12 + 6 * J + D + 3 * k;
        if (global_obs) { ok[0] = rel[0]; ok[1] = rel[1]; ok[2] = rel[2]; }
        else quat_rotate(h, rel, ok);
    }
}

/* envs/ig_parkour/mgdm_dm_util.py:462-519 with global_tar_root_h_obs=False (the only value its caller passes, :549).
 * one env, S target steps; out [S, 3+6+6J+3K].  global_obs (:476): nothing is rotated and the key-body offsets stay relative to
 * the TARGET root (the `+ root_pos_obs` of :497 sits inside the local branch). */
static void tar_obs1(int S, int J, int K, int global_obs, const float *ref_root_pos, const float *ref_root_rot, const float *tar_root_pos,
                     const float *tar_root_rot, const float *tar_joint_rot, const float *tar_key_pos, float *o) {
    float h[4];
    calc_heading_quat_inv(ref_root_rot, h);
    int W = 3 + 6 + 6 * J + 3 * K;
    for (int s = 0; s < S; ++s) {
        float *os = o + (size_t)s * W;
        const float *tp = tar_root_pos + 3 * s;
        float rp[3] = {tp[0] - ref_root_pos[0], tp[1] - ref_root_pos[1], tp[2] - ref_root_pos[2]};
        float rpo[3];
        float tr[4];
        if (global_obs) {
            for (int a = 0; a < 3; ++a) rpo[a] = rp[a];
            for (int a = 0; a < 4; ++a) tr[a] = tar_root_rot[4 * s + a];
        } else {
            quat_rotate(h, rp, rpo);
            quat_mul(h, tar_root_rot + 4 * s, tr);
        }
        os[0] = rpo[0]; os[1] = rpo[1]; os[2] = rpo[2];
        quat_to_tan_norm(tr, os + 3);
        for (int j = 0; j < J; ++j) quat_to_tan_norm(tar_joint_rot + ((size_t)s * J + j) * 4, os + 9 + 6 * j);
        for (int k = 0; k < K; ++k) {
            const float *kp = tar_key_pos + ((size_t)s * K + k) * 3;
            float rel[3] = {kp[0] - tp[0], kp[1] - tp[1], kp[2] - tp[2]};
            float r[3];
            if (global_obs) { r[0] = rel[0]; r[1] = rel[1]; r[2] = rel[2]; }
            else {
                quat_rotate(h, rel, r);
                r[0] += rpo[0]; r[1] += rpo[1]; r[2] += rpo[2];
            }
            os[9 + 6 * J + 3 * k + 0] = r[0];
            os[9 + 6 * J + 3 * k + 1] = r[1];
            os[9 + 6 * J + 3 * k + 2] = r[2];
        }
    }
}

/*
 * Full observation row, IGParkourEnv._compute_obs envs/ig_parkour/ig_parkour_env.py:1054-1244, every switch of it:
 *   [root_h]? char_obs | tar_obs? | tar_contacts? | char_contacts? | hf | target_xy? | replan_t?
 * root_h: global_root_height_obs (ig_char_env.py:621-623).  tar_obs: enable_tar_obs (mgdm_dm_util.py:540).  tar_contacts:
 * use_contact_info AND enable_tar_obs, char_contacts: use_contact_info (:1177-1186).  target_xy (NULL = has_target_xy_obs off): the
 * offset to the target rotated by minus the heading with cos / sin (rotate_2d_vec torch_util.py:620-631), :1212-1223.  replan_t
 * (has_replan_t: enable_replan_timer_obs on an env with motion-generator rows): one clock for every env (:1068-1069,:1225-1231).
 * Inputs are the simulator state (char_*), the clip database and the per-env motion bookkeeping.
 * motion_xy_offset[N,2] = motion_offsets[motion_id, terrain_id] - env_offset[:,0:2]  (dm_env.py:604-615).
 */
void orc_compute_obs_ex(CHAR_ARGS, MLIB_ARGS, int N, int S, const float *tar_steps_dt, int K, const int64_t *key_body_ids,
                        const int64_t *motion_ids, const float *motion_times, const float *motion_xy_offset,
                        const float *char_root_pos, const float *char_root_rot, const float *char_root_vel,
                        const float *char_root_ang_vel, const float *char_dof_pos, const float *char_dof_vel,
                        const float *contact_forces, const float *ray_hfs, int P, float contact_eps, int global_obs,
                        int root_height_obs, int enable_tar_obs, int use_contact_info, const float *target_xy,
                        int has_replan_t, float replan_t, float *obs, int obs_dim) {
    orc_char_t c = mk_char(CHAR_PASS);
    MLIB_MAKE;
    int Wc = 12 + 6 * J + D + 3 * K;
    int Wt = 3 + 6 + 6 * J + 3 * K;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        float *o = obs + (size_t)e * obs_dim;
        float jrot[64 * 4], bpos[65 * 3], brot[65 * 4], key[16 * 3];
        dof_to_rot1(&c, char_dof_pos + (size_t)e * D, jrot);
        forward_kinematics1(&c, char_root_pos + 3 * e, char_root_rot + 4 * e, jrot, bpos, brot);
        for (int k = 0; k < K; ++k)
            for (int a = 0; a < 3; ++a) key[3 * k + a] = bpos[3 * key_body_ids[k] + a];
        if (root_height_obs) *o++ = char_root_pos[3 * e + 2];
        char_obs1(J, D, K, global_obs, char_root_pos + 3 * e, char_root_rot + 4 * e, char_root_vel + 3 * e,
                  char_root_ang_vel + 3 * e, jrot, char_dof_vel + (size_t)e * D, key, o);
        o += Wc;
        /* target frames: fetch_tar_obs_data mgdm_dm_util.py:279-302, dm_env.compute_tar_obs dm_env.py:686-718 */
        float trp[8 * 3], trr[8 * 4], tjr[8 * 64 * 4], tkey[8 * 16 * 3], tcon[8 * 65];
        if (enable_tar_obs) {
            for (int s = 0; s < S; ++s) {
                float rv[3], rav[3], dv[128];
                float t = motion_times[e] + tar_steps_dt[s];
                calc_motion_frame1(&ml, motion_ids[e], t, trp + 3 * s, trr + 4 * s, rv, rav, tjr + (size_t)s * J * 4, dv, tcon + s * B);
                trp[3 * s + 0] += motion_xy_offset[2 * e + 0];
                trp[3 * s + 1] += motion_xy_offset[2 * e + 1];
                forward_kinematics1(&c, trp + 3 * s, trr + 4 * s, tjr + (size_t)s * J * 4, bpos, brot);
                for (int k = 0; k < K; ++k)
                    for (int a = 0; a < 3; ++a) tkey[(s * K + k) * 3 + a] = bpos[3 * key_body_ids[k] + a];
            }
            tar_obs1(S, J, K, global_obs, char_root_pos + 3 * e, char_root_rot + 4 * e, trp, trr, tjr, tkey, o);
            o += S * Wt;
        }
        if (use_contact_info) {
            if (enable_tar_obs) {
                for (int s = 0; s < S; ++s)
                    for (int b = 0; b < B; ++b) o[s * B + b] = tcon[s * B + b];
                o += S * B;
            }
            for (int b = 0; b < B; ++b) { /* ig_parkour_env.py:841-848 */
                const float *f = contact_forces + ((size_t)e * B + b) * 3;
                float nrm = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
                o[b] = nrm > contact_eps ? 1.f : 0.f;
            }
            o += B;
        }
        for (int p = 0; p < P; ++p) o[p] = ray_hfs[(size_t)e * P + p];
        o += P;
        if (target_xy) {
            float ang = -calc_heading(char_root_rot + 4 * e);
            float x = target_xy[2 * e] - char_root_pos[3 * e], y = target_xy[2 * e + 1] - char_root_pos[3 * e + 1];
            float ca = cosf(ang), sa = sinf(ang);
            o[0] = x * ca - y * sa;
            o[1] = x * sa + y * ca;
            o += 2;
        }
        if (has_replan_t) *o++ = replan_t;
        if ((int)(o - (obs + (size_t)e * obs_dim)) != obs_dim) abort();          /* the caller's row width is the reference's */
    }
}

/* the default tracker configuration (every switch at the value the reference ships for stage 3) */
void orc_compute_obs(CHAR_ARGS, MLIB_ARGS, int N, int S, const float *tar_steps_dt, int K, const int64_t *key_body_ids,
                     const int64_t *motion_ids, const float *motion_times, const float *motion_xy_offset,
                     const float *char_root_pos, const float *char_root_rot, const float *char_root_vel,
                     const float *char_root_ang_vel, const float *char_dof_pos, const float *char_dof_vel,
                     const float *contact_forces, const float *ray_hfs, int P, float contact_eps, float *obs, int obs_dim) {
    orc_compute_obs_ex(CHAR_PASS, MLIB_PASS, N, S, tar_steps_dt, K, key_body_ids, motion_ids, motion_times, motion_xy_offset,
                       char_root_pos, char_root_rot, char_root_vel, char_root_ang_vel, char_dof_pos, char_dof_vel, contact_forces,
                       ray_hfs, P, contact_eps, 0, 0, 1, 1, NULL, 0, 0.f, obs, obs_dim);
}

/* ------------------------------------------------------------------ reward (a10) */

/* the three root quantities and the key-body offsets in the heading frame: convert_to_local mgdm_dm_util.py:305-325 */
static void to_local1(int K, float *root_rot, float *root_vel, float *root_ang_vel, float *key) {
    float h[4], q[4], v[3];
    calc_heading_quat_inv(root_rot, h);
    quat_mul(h, root_rot, q);
    for (int a = 0; a < 4; ++a) root_rot[a] = q[a];
    quat_rotate(h, root_vel, v);
    for (int a = 0; a < 3; ++a) root_vel[a] = v[a];
    quat_rotate(h, root_ang_vel, v);
    for (int a = 0; a < 3; ++a) root_ang_vel[a] = v[a];
    for (int k = 0; k < K; ++k) {
        quat_rotate(h, key + 3 * k, v);
        for (int a = 0; a < 3; ++a) key[3 * k + a] = v[a];
    }
}

/* IGParkourEnv._update_reward ig_parkour_env.py:1275-1404, every switch of it: compute_deepmimic_reward mgdm_dm_util.py:327-390
 * (track_root: the horizontal root error is dropped :352-353 and root rotation / velocities / key-body offsets of both sides are
 * compared in their own heading frames :364-366; track_root_h: the vertical root error is dropped :355-356), the contact term
 * mgdm_dm_util.py:555-576 if use_contact_info (:1324-1339), the target-location task reward :1347-1388 (target_xy NULL: the target is
 * the character's own position, what the reference holds before a target is set) and the product rule for rel_task_w > 0 (:1399-1404).
 * terms[N,9] = pose, vel, root_pose, root_vel, key_pos, contact_penalty, task_r1, task_r2, total_task_r */
void orc_compute_reward_ex(CHAR_ARGS, int N, int K, const int64_t *key_body_ids, const float *char_root_pos,
                           const float *char_root_rot, const float *char_root_vel, const float *char_root_ang_vel,
                           const float *char_dof_pos, const float *char_dof_vel, const float *char_body_pos,
                           const float *ref_root_pos, const float *ref_root_rot, const float *ref_root_vel,
                           const float *ref_root_ang_vel, const float *ref_joint_rot, const float *ref_dof_vel,
                           const float *ref_body_pos, const float *ref_contacts, const float *contact_forces,
                           const float *joint_err_w, const float *dof_err_w, const float *contact_w, const float *w5,
                           float rel_dm_w, int track_root, int track_root_h, int use_contact_info, const float *target_xy,
                           float task1_w, float task2_w, float target_radius, float rel_task_w, float *reward, float *terms) {
    orc_char_t c = mk_char(CHAR_PASS);
    int D = char_dof_size(&c), J = nb - 1, B = nb;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        float jrot[64 * 4];
        dof_to_rot1(&c, char_dof_pos + (size_t)e * D, jrot);
        float pose_err = 0.f;
        for (int j = 0; j < J; ++j) {
            float d = quat_diff_angle(jrot + 4 * j, ref_joint_rot + ((size_t)e * J + j) * 4);
            pose_err += joint_err_w[j] * d * d;
        }
        float vel_err = 0.f;
        for (int d = 0; d < D; ++d) {
            float dv = ref_dof_vel[(size_t)e * D + d] - char_dof_vel[(size_t)e * D + d];
            vel_err += dof_err_w[d] * dv * dv;
        }
        const float *rp = char_root_pos + 3 * e, *tp = ref_root_pos + 3 * e;
        float root_pos_err = 0.f;
        for (int a = 0; a < 3; ++a) {
            float d = tp[a] - rp[a];
            if ((a < 2 && !track_root) || (a == 2 && !track_root_h)) d = 0.f;
            root_pos_err += d * d;
        }
        float rq[4], rv[3], rw[3], tq[4], tv[3], tw[3], ckey[16 * 3], tkey[16 * 3];
        for (int a = 0; a < 4; ++a) { rq[a] = char_root_rot[4 * e + a]; tq[a] = ref_root_rot[4 * e + a]; }
        for (int a = 0; a < 3; ++a) {
            rv[a] = char_root_vel[3 * e + a]; rw[a] = char_root_ang_vel[3 * e + a];
            tv[a] = ref_root_vel[3 * e + a]; tw[a] = ref_root_ang_vel[3 * e + a];
        }
        for (int k = 0; k < K; ++k) {
            int64_t b = key_body_ids[k];
            for (int a = 0; a < 3; ++a) {
                ckey[3 * k + a] = char_body_pos[((size_t)e * B + b) * 3 + a] - rp[a];
                tkey[3 * k + a] = ref_body_pos[((size_t)e * B + b) * 3 + a] - tp[a];
            }
        }
        if (!track_root) {
            to_local1(K, rq, rv, rw, ckey);
            to_local1(K, tq, tv, tw, tkey);
        }
        float rre = quat_diff_angle(rq, tq);
        rre *= rre;
        float rve = 0.f, rave = 0.f;
        for (int a = 0; a < 3; ++a) {
            float d = tv[a] - rv[a];
            rve += d * d;
            float d2 = tw[a] - rw[a];
            rave += d2 * d2;
        }
        float kpe = 0.f;
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
            for (int a = 0; a < 3; ++a) {
                float d = tkey[3 * k + a] - ckey[3 * k + a];
                s += d * d;
            }
            kpe += s;
        }
        float pose_r = expf(-0.25f * pose_err);
        float vel_r = expf(-0.01f * vel_err);
        float root_pose_r = expf(-5.0f * (root_pos_err + 0.1f * rre));
        float root_vel_r = expf(-1.0f * (rve + 0.1f * rave));
        float key_r = expf(-10.0f * kpe);
        float cp = 0.f;
        for (int b = 0; b < B; ++b) {
            const float *f = contact_forces + ((size_t)e * B + b) * 3;
            float fn = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
            if (fn > 1.0f) fn = 1.0f;
            float tc = ref_contacts[(size_t)e * B + b];
            float r = -(1.0f - tc) * fn;
            r += tc * fn;
            cp += contact_w[b] * r;
        }
        cp /= (float)B;
        float dm = w5[0] * pose_r + w5[1] * vel_r + w5[2] * root_pose_r + w5[3] * root_vel_r + w5[4] * key_r;
        if (use_contact_info) dm += cp;
        /* task reward: reach the target location at >= 2 m/s facing it (:1347-1388) */
        float dx = (target_xy ? target_xy[2 * e] : rp[0]) - rp[0], dy = (target_xy ? target_xy[2 * e + 1] : rp[1]) - rp[1];
        float err = dx * dx + dy * dy;
        float r1 = expf(-0.075f * err);
        float len = sqrtf(err);
        float ux = len > 0.01f ? dx / len : 0.f, uy = len > 0.01f ? dy / len : 0.f;
        float mv = 2.0f - (ux * char_root_vel[3 * e] + uy * char_root_vel[3 * e + 1]);
        if (mv < 0.f) mv = 0.f;
        float min_vel_r = expf(-(mv * mv));
        float hd = calc_heading(char_root_rot + 4 * e);
        float he = 1.0f - (ux * cosf(hd) + uy * sinf(hd));
        if (he < 0.f) he = 0.f;
        float r2 = min_vel_r * expf(-(he * he));
        float task = task1_w * r1 + task2_w * r2;
        if (err < target_radius * target_radius) task = 1.0f;
        reward[e] = rel_task_w > 0.f ? dm * task : rel_dm_w * dm;
        float *t9 = terms + 9 * (size_t)e;
        t9[0] = pose_r; t9[1] = vel_r; t9[2] = root_pose_r; t9[3] = root_vel_r; t9[4] = key_r; t9[5] = cp;
        t9[6] = r1; t9[7] = r2; t9[8] = task;
    }
}

/* the default tracker configuration; terms[N,6] = pose, vel, root_pose, root_vel, key_pos, contact_penalty */
void orc_compute_reward(CHAR_ARGS, int N, int K, const int64_t *key_body_ids, const float *char_root_pos,
                        const float *char_root_rot, const float *char_root_vel, const float *char_root_ang_vel,
                        const float *char_dof_pos, const float *char_dof_vel, const float *char_body_pos,
                        const float *ref_root_pos, const float *ref_root_rot, const float *ref_root_vel,
                        const float *ref_root_ang_vel, const float *ref_joint_rot, const float *ref_dof_vel,
                        const float *ref_body_pos, const float *ref_contacts, const float *contact_forces,
                        const float *joint_err_w, const float *dof_err_w, const float *contact_w, const float *w5,
                        float rel_dm_w, float *reward, float *terms) {
    float *t9 = (float *)malloc(sizeof(float) * 9 * (size_t)(N > 0 ? N : 1));
    orc_compute_reward_ex(CHAR_PASS, N, K, key_body_ids, char_root_pos, char_root_rot, char_root_vel, char_root_ang_vel, char_dof_pos,
                          char_dof_vel, char_body_pos, ref_root_pos, ref_root_rot, ref_root_vel, ref_root_ang_vel, ref_joint_rot,
                          ref_dof_vel, ref_body_pos, ref_contacts, contact_forces, joint_err_w, dof_err_w, contact_w, w5, rel_dm_w,
                          1, 1, 1, NULL, 0.7f, 0.3f, 1.0f, 0.f, reward, t9);
    for (int e = 0; e < N; ++e)
        for (int k = 0; k < 6; ++k) terms[6 * (size_t)e + k] = t9[9 * (size_t)e + k];
    free(t9);
}

/* ------------------------------------------------------------------ termination (a11) */

/* mgdm_dm_util.py:205-230,392-460 + dm_env.py:746-783.  contact_body_mask[B]=1 for bodies allowed to touch.
 * fail_rates[M] is updated sequentially in env order as the reference's Python loop does. */
void orc_update_done(int N, int B, int M, const float *time_buf, float ep_len, const float *char_root_rot,
                     const float *body_pos, const float *ref_root_rot, const float *ref_body_pos,
                     const float *contact_forces, int n_contact_bodies, const int *contact_body_mask,
                     const float *env_offsets, const float *hf, int dim_x, int dim_y, float min_x, float min_y, float dx,
                     float dy, float termination_height, int pose_termination, const float *pose_termination_dist,
                     int enable_early_termination, int track_root, float root_pos_term_dist, float root_rot_term_angle,
                     const int64_t *motion_ids, const float *motion_times, const float *motion_len,
                     const int *motion_loop_mode, float ema_w, int *done_pre, int *done, float *fail_rates) {
    for (int e = 0; e < N; ++e) {
        int d = 0;
        if (time_buf[e] >= ep_len) d = 3;
        if (enable_early_termination) {
            int failed = 0;
            if (n_contact_bodies > 0) {
                int fall_contact = 0, fall_height = 0;
                for (int b = 0; b < B; ++b) {
                    if (contact_body_mask[b]) continue;
                    const float *f = contact_forces + ((size_t)e * B + b) * 3;
                    if (fabsf(f[0]) > 0.1f || fabsf(f[1]) > 0.1f || fabsf(f[2]) > 0.1f) fall_contact = 1;
                    const float *bp = body_pos + ((size_t)e * B + b) * 3;
                    float th = hf_lookup(hf, dim_x, dim_y, min_x, min_y, dx, dy, bp[0] + env_offsets[3 * e], bp[1] + env_offsets[3 * e + 1]) + termination_height;
                    if (bp[2] < th) fall_height = 1;
                }
                failed = fall_contact && fall_height;
            }
            if (pose_termination) {
                const float *r0 = body_pos + (size_t)e * B * 3, *t0 = ref_body_pos + (size_t)e * B * 3;
                int pose_fail = 0;
                for (int b = 1; b < B; ++b) {
                    float s = 0.f;
                    for (int a = 0; a < 3; ++a) {
                        float bp = r0[3 * b + a] - r0[a];
                        float tb = t0[3 * b + a] - t0[a];
                        float df = tb - bp;
                        s += df * df;
                    }
                    float lim = pose_termination_dist[b - 1];
                    if (s > lim * lim) pose_fail = 1;
                }
                if (track_root) {
                    float s = 0.f;
                    for (int a = 0; a < 3; ++a) { float df = r0[a] - t0[a]; s += df * df; }
                    if (s > root_pos_term_dist * root_pos_term_dist) pose_fail = 1;
                    float ang = quat_diff_angle(char_root_rot + 4 * e, ref_root_rot + 4 * e);
                    if (fabsf(ang) > root_rot_term_angle) pose_fail = 1;
                }
                failed = failed || pose_fail;
            }
            if (!(time_buf[e] > 1e-5f)) failed = 0;
            if (failed) d = 1;
        }
        done_pre[e] = d;
        int64_t m = motion_ids[e];
        int motion_end = (motion_times[e] >= motion_len[m]) && (motion_loop_mode[m] != 1);
        if (d != 0 || motion_end) {
            if (d == 1) fail_rates[m] = fail_rates[m] * (1.0f - ema_w) + ema_w;
            else fail_rates[m] = fail_rates[m] * (1.0f - ema_w);
        }
        if (motion_end) d = 1;
        done[e] = d;
        (void)M;
    }
}

/* mgdm_dm_util.py:578-611; out [N,7] */
void orc_tracking_error(int N, int B, int D, const float *root_pos, const float *root_rot, const float *body_rot,
                        const float *body_pos, const float *tar_root_pos, const float *tar_root_rot,
                        const float *tar_body_rot, const float *tar_body_pos, const float *root_vel,
                        const float *root_ang_vel, const float *dof_vel, const float *tar_root_vel,
                        const float *tar_root_ang_vel, const float *tar_dof_vel, float *out) {
    for (int e = 0; e < N; ++e) {
        float pose = 0.f, bpe = 0.f;
        for (int b = 0; b < B; ++b) {
            pose += fabsf(quat_diff_angle(body_rot + ((size_t)e * B + b) * 4, tar_body_rot + ((size_t)e * B + b) * 4));
            float s = 0.f;
            for (int a = 0; a < 3; ++a) {
                float x = body_pos[((size_t)e * B + b) * 3 + a] - root_pos[3 * e + a];
                float y = tar_body_pos[((size_t)e * B + b) * 3 + a] - tar_root_pos[3 * e + a];
                s += (y - x) * (y - x);
            }
            bpe += sqrtf(s);
        }
        float rpd = 0.f, rv = 0.f, rav = 0.f, dve = 0.f;
        for (int a = 0; a < 3; ++a) {
            float d = tar_root_pos[3 * e + a] - root_pos[3 * e + a];
            rpd += d * d;
            rv += fabsf(tar_root_vel[3 * e + a] - root_vel[3 * e + a]);
            rav += fabsf(tar_root_ang_vel[3 * e + a] - root_ang_vel[3 * e + a]);
        }
        for (int d = 0; d < D; ++d) dve += fabsf(tar_dof_vel[(size_t)e * D + d] - dof_vel[(size_t)e * D + d]);
        float *o = out + 7 * (size_t)e;
        o[0] = sqrtf(rpd);
        o[1] = fabsf(quat_diff_angle(root_rot + 4 * e, tar_root_rot + 4 * e));
        o[2] = bpe / (float)B;
        o[3] = pose / (float)B;
        o[4] = dve / (float)D;
        o[5] = rv / 3.f;
        o[6] = rav / 3.f;
    }
}

/* ------------------------------------------------------------------ TD(lambda) / advantage (a16) */

/* learning/rl_util.py:6-29 ; r, next_vals, ret [T,N] time-major; done int32 */
void orc_td_lambda_return(int T, int N, const float *r, const float *next_vals, const int *done, float discount,
                          float td_lambda, float *ret) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        ret[(size_t)(T - 1) * N + e] = r[(size_t)(T - 1) * N + e] + discount * next_vals[(size_t)(T - 1) * N + e];
        for (int i = T - 2; i >= 0; --i) {
            float reset = done[(size_t)i * N + e] != 0 ? 1.f : 0.f;
            float lam = td_lambda * (1.0f - reset);
            float nv = next_vals[(size_t)i * N + e];
            float nr = ret[(size_t)(i + 1) * N + e];
            ret[(size_t)i * N + e] = r[(size_t)i * N + e] + discount * ((1.0f - lam) * nv + lam * nr);
        }
    }
}

/* learning/dm_ppo_agent.py:393-403 ; torch.std_mean = unbiased std.  n = T*N flat */
void orc_adv_normalize(int n, const float *ret, const float *vals, const float *rand_mask, float clip, float *norm_adv,
                       float *mean_out, float *std_out) {
    double s = 0.0;
    int64_t cnt = 0;
    for (int i = 0; i < n; ++i)
        if (rand_mask[i] == 1.0f) { s += (double)(ret[i] - vals[i]); ++cnt; }
    double mean = cnt ? s / (double)cnt : 0.0;
    double ss = 0.0;
    for (int i = 0; i < n; ++i)
        if (rand_mask[i] == 1.0f) { double d = (double)(ret[i] - vals[i]) - mean; ss += d * d; }
    double std = cnt > 1 ? sqrt(ss / (double)(cnt - 1)) : 0.0;
    float meanf = (float)mean, stdf = (float)std;
    float den = stdf < 1e-5f ? 1e-5f : stdf;
    for (int i = 0; i < n; ++i) {
        float a = ((ret[i] - vals[i]) - meanf) / den;
        if (a < -clip) a = -clip;
        if (a > clip) a = clip;
        norm_adv[i] = a;
    }
    *mean_out = meanf;
    *std_out = stdf;
}

/* ------------------------------------------------------------------ whole kinematic step (cpu_baseline) */

/* K3 + shift + K2 + K4 for the reference character (dm_env.py:570-595): writes ref_* rows for N envs */
void orc_update_ref_motion(CHAR_ARGS, MLIB_ARGS, int N, const int64_t *motion_ids, const float *motion_times,
                           const float *motion_xy_offset, float *ref_root_pos, float *ref_root_rot, float *ref_root_vel,
                           float *ref_root_ang_vel, float *ref_joint_rot, float *ref_dof_vel, float *ref_contacts,
                           float *ref_body_pos, float *ref_dof_pos) {
    orc_char_t c = mk_char(CHAR_PASS);
    MLIB_MAKE;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < N; ++e) {
        float brot[65 * 4];
        calc_motion_frame1(&ml, motion_ids[e], motion_times[e], ref_root_pos + 3 * e, ref_root_rot + 4 * e, ref_root_vel + 3 * e,
                           ref_root_ang_vel + 3 * e, ref_joint_rot + (size_t)e * J * 4, ref_dof_vel + (size_t)e * D,
                           ref_contacts + (size_t)e * B);
        ref_root_pos[3 * e + 0] += motion_xy_offset[2 * e + 0];
        ref_root_pos[3 * e + 1] += motion_xy_offset[2 * e + 1];
        forward_kinematics1(&c, ref_root_pos + 3 * e, ref_root_rot + 4 * e, ref_joint_rot + (size_t)e * J * 4,
                            ref_body_pos + (size_t)e * nb * 3, brot);
        rot_to_dof1(&c, ref_joint_rot + (size_t)e * J * 4, ref_dof_pos + (size_t)e * D, D);
    }
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
