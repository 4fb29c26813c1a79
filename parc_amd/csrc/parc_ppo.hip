// PPO surrogate + critic loss of the tracker agent with its gradient in one pass (gfx950).
// Restates PPOAgent._compute_loss / DMPPOAgent (learning/ppo_agent.py:186-330, learning/dm_ppo_agent.py) for a diagonal
// Gaussian policy with state-independent log-std: the ~120 small elementwise / reduction launches of the autograd
// graph between the two MLP outputs and the scalar loss become three launches with a deterministic reduction order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/parc_hip.h"

#define PPO_THREADS 256
#define PPO_MAX_A 64
#define PPO_NSCAL 8      // cnt_raw, critic_sum, surr_sum, clip_sum, ratio_sum, viol_sum, reg_sum, (spare)
#define PPO_W (PPO_NSCAL + PPO_MAX_A)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// pass 1: one thread per sample.  Unscaled gradients (the 1/cnt and gate factors are only known after the reduction).
__global__ __launch_bounds__(PPO_THREADS) void ppo_sample_kernel(int B, int A, const float *__restrict__ mean, const float *__restrict__ logstd,
                                                                 const float *__restrict__ norm_a, const float *__restrict__ old_logp,
                                                                 const float *__restrict__ adv, const float *__restrict__ mask,
                                                                 const float *__restrict__ pred, const float *__restrict__ tar_val,
                                                                 parc_ppo_cfg_t cfg, float *g_mean, float *g_pred, float *ws, int a_stride,
                                                                 int s_stride) {
    __shared__ float s_istd[PPO_MAX_A];
    __shared__ float s_part[PPO_THREADS / 64][PPO_W];
    const int tid = threadIdx.x, i = blockIdx.x * PPO_THREADS + tid;
    float sum_logstd = 0.f;
    if (tid < A) s_istd[tid] = __expf(-logstd[tid]);       // std^-1 = exp(-logstd)
    __syncthreads();
    for (int j = 0; j < A; ++j) sum_logstd += logstd[j];
    const bool live = i < B;
    const int ii = live ? i : 0;
    // (a_stride / s_stride: row stride of norm_a and element stride of the four per-sample scalars; A and 1 for separate dense arrays,
    //  the record width for the packed layout of parc_ppo_loss_packed)
    const size_t si = (size_t)ii * s_stride;
    const float m = live ? (mask[si] == 1.0f ? 1.f : 0.f) : 0.f;
    const float *mu = mean + (size_t)ii * A, *ac = norm_a + (size_t)ii * a_stride;
    float zz = 0.f, viol = 0.f, reg = 0.f;
    for (int j = 0; j < A; ++j) {
        float mj = mu[j];
        float z = (ac[j] - mj) * s_istd[j];
        zz = fmaf(z, z, zz);
        float vmin = fminf(mj + 1.0f, 0.f), vmax = fmaxf(mj - 1.0f, 0.f);
        viol += vmin * vmin + vmax * vmax;
        reg = fmaf(mj, mj, reg);
    }
    const float logp = -0.5f * zz + (-0.5f * (float)A * 1.8378770664093453f - sum_logstd);     // log(2 pi)
    const float ratio = __expf(logp - old_logp[si]);
    const float ad = adv[si];
    const float lo = 1.0f - cfg.clip_ratio, hi = 1.0f + cfg.clip_ratio;
    const float rc = fminf(fmaxf(ratio, lo), hi);
    const float l0 = ad * ratio, l1 = ad * rc;
    const float surr = fminf(l0, l1);
    // d surr / d ratio: torch.minimum sends the gradient to l0 when l0 <= l1, else to l1, whose clamp passes it inside [lo, hi]
    const float dsurr = (l0 <= l1) ? ad : ((ratio >= lo && ratio <= hi) ? ad : 0.f);
    const float c = -m * dsurr * ratio;                       // d(-sum m surr)/d logp_i   (x 1/cnt later)
    const float diff = tar_val[si] - pred[ii];
    if (live) {
        g_pred[i] = cfg.critic_l1 ? (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) : -2.0f * diff;
        float *gm = g_mean + (size_t)i * A;
        for (int j = 0; j < A; ++j) {
            float mj = mu[j];
            float z = (ac[j] - mj) * s_istd[j];
            float vmin = fminf(mj + 1.0f, 0.f), vmax = fmaxf(mj - 1.0f, 0.f);
            gm[j] = c * z * s_istd[j] + m * (cfg.bound_w * 2.0f * (vmin + vmax) + cfg.reg_w * 2.0f * mj);
        }
    }
    // block partials, fixed order: lanes -> waves -> block
    float sc[PPO_NSCAL];
    sc[0] = m;
    sc[1] = live ? (cfg.critic_l1 ? fabsf(diff) : diff * diff) : 0.f;
    sc[2] = m * surr;
    sc[3] = m * (fabsf(ratio - 1.0f) > cfg.clip_ratio ? 1.f : 0.f);
    sc[4] = m * ratio;
    sc[5] = m * viol;
    sc[6] = m * reg;
    sc[7] = 0.f;
    const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
    for (int k = 0; k < PPO_NSCAL; ++k) {
        float v = wave_sum(sc[k]);
        if (ln == 0) s_part[wv][k] = v;
    }
    for (int j = 0; j < A; ++j) {
        float z = live ? (ac[j] - mu[j]) * s_istd[j] : 0.f;
        float v = wave_sum(c * (z * z - 1.0f));               // d logp / d logstd_j = z^2 - 1
        if (ln == 0) s_part[wv][PPO_NSCAL + j] = v;
    }
    __syncthreads();
    if (tid < PPO_NSCAL + A) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PPO_THREADS / 64; ++w) v += s_part[w][tid];
        ws[(size_t)blockIdx.x * PPO_W + tid] = v;
    }
}

// pass 1 for A <= 32 (the humanoid has 28 actuated dofs): 32 lanes per sample, lane j = action dimension j, so the [B, A] rows of mean /
// action / gradient are read and written coalesced (one thread per sample walks them with a 112-byte stride between lanes: 22 us for
// 16384 samples).  A block takes 64 samples in 8 passes; the per-sample scalars accumulate on lane 0 of each group and the log-std
// gradient of dimension j on lane j, folded over the 8 groups through LDS at the end (fixed order).
#define PPO32_SPB 64
__global__ __launch_bounds__(256) void ppo_sample32_kernel(int B, int A, const float *__restrict__ mean, const float *__restrict__ logstd,
                                                           const float *__restrict__ norm_a, const float *__restrict__ old_logp,
                                                           const float *__restrict__ adv, const float *__restrict__ mask,
                                                           const float *__restrict__ pred, const float *__restrict__ tar_val, parc_ppo_cfg_t cfg,
                                                           float *g_mean, float *g_pred, float *ws, int a_stride, int s_stride) {
    __shared__ float s_part[8][PPO_W];
    const int tid = threadIdx.x, g = tid >> 5, j = tid & 31;
    const bool vj = j < A;
    const float ls = vj ? logstd[j] : 0.f;
    const float istd = __expf(-ls);
    float sum_logstd = ls;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) sum_logstd += __shfl_xor(sum_logstd, o, 32);
    float acc[PPO_NSCAL];
#pragma unroll
    for (int k = 0; k < PPO_NSCAL; ++k) acc[k] = 0.f;
    float acc_ls = 0.f;
    const float lo = 1.0f - cfg.clip_ratio, hi = 1.0f + cfg.clip_ratio;
#pragma unroll 2
    for (int it = 0; it < PPO32_SPB / 8; ++it) {
        const int i = blockIdx.x * PPO32_SPB + it * 8 + g;
        const bool live = i < B;
        const int ii = live ? i : 0;
        const size_t si = (size_t)ii * s_stride;
        const float m = live ? (mask[si] == 1.0f ? 1.f : 0.f) : 0.f;
        const float mj = vj ? mean[(size_t)ii * A + j] : 0.f;
        const float aj = vj ? norm_a[(size_t)ii * a_stride + j] : 0.f;
        const float z = (aj - mj) * istd;
        const float vmin = fminf(mj + 1.0f, 0.f), vmax = fmaxf(mj - 1.0f, 0.f);
        float zz = vj ? z * z : 0.f, viol = vj ? vmin * vmin + vmax * vmax : 0.f, reg = vj ? mj * mj : 0.f;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            zz += __shfl_xor(zz, o, 32);
            viol += __shfl_xor(viol, o, 32);
            reg += __shfl_xor(reg, o, 32);
        }
        const float logp = -0.5f * zz + (-0.5f * (float)A * 1.8378770664093453f - sum_logstd);
        const float ratio = __expf(logp - old_logp[si]);
        const float ad = adv[si];
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float l0 = ad * ratio, l1 = ad * rc;
        const float surr = fminf(l0, l1);
        const float dsurr = (l0 <= l1) ? ad : ((ratio >= lo && ratio <= hi) ? ad : 0.f);
        const float c = -m * dsurr * ratio;
        const float diff = tar_val[si] - pred[ii];
        if (live) {
            if (j == 0) g_pred[i] = cfg.critic_l1 ? (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) : -2.0f * diff;
            if (vj) g_mean[(size_t)i * A + j] = c * z * istd + m * (cfg.bound_w * 2.0f * (vmin + vmax) + cfg.reg_w * 2.0f * mj);
        }
        // (every lane of the group holds the same scalars; lane 0's copy is the one that is kept)
        acc[0] += m;
        acc[1] += live ? (cfg.critic_l1 ? fabsf(diff) : diff * diff) : 0.f;
        acc[2] += m * surr;
        acc[3] += m * (fabsf(ratio - 1.0f) > cfg.clip_ratio ? 1.f : 0.f);
        acc[4] += m * ratio;
        acc[5] += m * viol;
        acc[6] += m * reg;
        acc_ls += (live && vj) ? c * (z * z - 1.0f) : 0.f;
    }
    if (j == 0) {
#pragma unroll
        for (int k = 0; k < PPO_NSCAL; ++k) s_part[g][k] = acc[k];
    }
    if (vj) s_part[g][PPO_NSCAL + j] = acc_ls;
    __syncthreads();
    if (tid < PPO_NSCAL + A) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += s_part[w][tid];
        ws[(size_t)blockIdx.x * PPO_W + tid] = v;
    }
}

// pass 2: one small block reduces the block partials in order and produces the scalars, the log-std gradient and the scale factors
#define PPO_RED_THREADS (4 * PPO_W)
__global__ __launch_bounds__(PPO_RED_THREADS) void ppo_reduce_kernel(int B, int A, int nblk, const float *__restrict__ logstd, parc_ppo_cfg_t cfg,
                                                                    const float *__restrict__ ws, float *g_logstd, float *out) {
    __shared__ float s[PPO_W];
    __shared__ double s4[PPO_W][4];
    const int tid = threadIdx.x;
    {
        // value v of the block partials: four threads add a quarter of the blocks each, in block order, their sums are then added in
        // quarter order (fixed summation order, deterministic); unrolled so that 16 loads are in flight instead of one dependent load
        // per step (that loop was most of the kernel's 26 us)
        const int v = tid >> 2, part = tid & 3;
        const int per = (nblk + 3) >> 2, b0 = part * per, b1 = min(b0 + per, nblk);
        double acc = 0.0;
        if (v < PPO_NSCAL + A) {
#pragma unroll 16
            for (int b = b0; b < b1; ++b) acc += (double)ws[(size_t)b * PPO_W + v];
        }
        s4[v][part] = acc;
    }
    __syncthreads();
    if (tid < PPO_NSCAL + A) s[tid] = (float)((s4[tid][0] + s4[tid][1]) + (s4[tid][2] + s4[tid][3]));
    __syncthreads();
    const float msum = s[0];
    // no random-action sample in the batch: the reference takes means over an empty selection, which are NaN, and its NaN trap
    // then stops the run (ppo_agent.py:242-252).  Dividing by the raw count reproduces that (0/0) instead of hiding it.
    const float cnt = msum;
    const float critic_loss = s[1] / (float)B;
    float sum_logstd = 0.f;
    for (int j = 0; j < A; ++j) sum_logstd += logstd[j];
    const float ent = (sum_logstd + 0.5f * (float)A * 2.8378770664093453f) * (msum / cnt);     // log(2 pi e)
    const float abl = s[5] / cnt, regl = s[6] / cnt;
    float actor_loss = -s[2] / cnt;
    if (cfg.bound_w != 0.f) actor_loss += cfg.bound_w * abl;
    if (cfg.entropy_w != 0.f) actor_loss -= cfg.entropy_w * ent;
    if (cfg.reg_w != 0.f) actor_loss += cfg.reg_w * regl;
    // "LARGE CRITIC LOSS" guard (ppo_agent.py:225-238): the actor term gives no gradient while the critic is off
    const float gate = critic_loss > cfg.large_critic_loss ? 0.f : 1.f;
    if (tid < A) g_logstd[tid] = gate * (s[PPO_NSCAL + tid] / cnt - cfg.entropy_w * (msum / cnt));
    if (tid == 0) {
        out[0] = actor_loss + cfg.critic_w * critic_loss;
        out[1] = critic_loss;
        out[2] = actor_loss;
        out[3] = s[3] / cnt;
        out[4] = s[4] / cnt;
        out[5] = abl;
        out[6] = ent;
        out[7] = regl;
        out[8] = cnt;
        out[9] = gate / cnt;                         // scale of g_mean
        out[10] = cfg.critic_w / (float)B;           // scale of g_pred
    }
}

// pass 3: apply the scale factors
__global__ __launch_bounds__(256) void ppo_scale_kernel(int n_mean, int n_pred, const float *__restrict__ out, float *g_mean, float *g_pred) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float sa = out[9], sp = out[10];
    if (i < n_mean) g_mean[i] *= sa;
    if (i < n_pred) g_pred[i] *= sp;
}

static int ppo_loss_impl(void *stream, int B, int A, const float *mean, const float *logstd, const float *norm_a, const float *old_logp,
                         const float *adv, const float *mask, const float *pred, const float *tar_val, parc_ppo_cfg_t cfg, float *g_mean,
                         float *g_logstd, float *g_pred, float *out, float *workspace, int a_stride, int s_stride);

extern "C" int parc_ppo_loss(void *stream, int B, int A, const float *mean, const float *logstd, const float *norm_a, const float *old_logp,
                             const float *adv, const float *mask, const float *pred, const float *tar_val, parc_ppo_cfg_t cfg, float *g_mean,
                             float *g_logstd, float *g_pred, float *out, float *workspace) {
    return ppo_loss_impl(stream, B, A, mean, logstd, norm_a, old_logp, adv, mask, pred, tar_val, cfg, g_mean, g_logstd, g_pred, out, workspace, A, 1);
}

// the same loss on PACKED per-sample records rec[B, rec_stride] = [norm_action (A) | a_logp | adv | rand_action_mask | tar_val | pad]:
// what one row gather of a minibatch delivers (the update phase gathers two arrays per minibatch instead of six)
extern "C" int parc_ppo_loss_packed(void *stream, int B, int A, const float *mean, const float *logstd, const float *rec, int rec_stride,
                                    const float *pred, parc_ppo_cfg_t cfg, float *g_mean, float *g_logstd, float *g_pred, float *out,
                                    float *workspace) {
    if (!rec || rec_stride < A + 4) return PARC_EINVAL;
    return ppo_loss_impl(stream, B, A, mean, logstd, rec, rec + A, rec + A + 1, rec + A + 2, pred, rec + A + 3, cfg, g_mean, g_logstd, g_pred, out,
                         workspace, rec_stride, rec_stride);
}

static int ppo_loss_impl(void *stream, int B, int A, const float *mean, const float *logstd, const float *norm_a, const float *old_logp,
                         const float *adv, const float *mask, const float *pred, const float *tar_val, parc_ppo_cfg_t cfg, float *g_mean,
                         float *g_logstd, float *g_pred, float *out, float *workspace, int a_stride, int s_stride) {
    if (B <= 0 || A <= 0 || A > PPO_MAX_A) return PARC_EINVAL;
    int nblk = (B + PPO_THREADS - 1) / PPO_THREADS;
    hipStream_t st = (hipStream_t)stream;
    if (A <= 32) {
        nblk = (B + PPO32_SPB - 1) / PPO32_SPB;
        hipLaunchKernelGGL(ppo_sample32_kernel, dim3(nblk), dim3(256), 0, st, B, A, mean, logstd, norm_a, old_logp, adv, mask, pred, tar_val, cfg,
                           g_mean, g_pred, workspace, a_stride, s_stride);
    } else {
        hipLaunchKernelGGL(ppo_sample_kernel, dim3(nblk), dim3(PPO_THREADS), 0, st, B, A, mean, logstd, norm_a, old_logp, adv, mask, pred, tar_val,
                           cfg, g_mean, g_pred, workspace, a_stride, s_stride);
    }
    hipLaunchKernelGGL(ppo_reduce_kernel, dim3(1), dim3(PPO_RED_THREADS), 0, st, B, A, nblk, logstd, cfg, workspace, g_logstd, out);
    const int n = B * A;
    hipLaunchKernelGGL(ppo_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, B, out, g_mean, g_pred);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

extern "C" int parc_ppo_workspace_floats(int B) { return ((B + PPO32_SPB - 1) / PPO32_SPB) * PPO_W; }      // (the finer of the two block sizes)

// =============================================================================================
// K15 experience record: ExperienceBuffer.record (learning/experience_buffer.py:55-59) for a whole group of named buffers in one
// launch: field f copies its [N, row] source into row `*head` of its time-major [T, N, row] buffer.  head is a DEVICE scalar
// so the launch can sit inside the captured rollout graph.
// =============================================================================================
#define PARC_RECORD_MAX_FIELDS 12
struct record_fields_t {
    parc_record_field_t f[PARC_RECORD_MAX_FIELDS];
};

__global__ __launch_bounds__(256) void record_step_kernel(int n_envs, const int64_t *__restrict__ head, record_fields_t fields) {
    const parc_record_field_t f = fields.f[blockIdx.y];
    const size_t total = (size_t)n_envs * (size_t)f.row_bytes;            // bytes of one time row
    const size_t h = (size_t)head[0];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
    if (f.convert == 1) {                                                 // int64 source -> int32 buffer (ep_num)
        const int64_t *s = (const int64_t *)f.src;
        int32_t *d = (int32_t *)f.dst + h * (total / 8);
        for (size_t i = tid; i < total / 8; i += nthr) d[i] = (int32_t)s[i];
        return;
    }
    if (f.convert == 2) {                                                 // one 4-byte value for every env (the plan clock: replan_timer)
        const uint32_t v = *(const uint32_t *)f.src;
        uint32_t *d = (uint32_t *)f.dst + h * (total / 4);
        for (size_t i = tid; i < total / 4; i += nthr) d[i] = v;
        return;
    }
    char *d = (char *)f.dst + h * total;
    const char *s = (const char *)f.src;
    if (((total | (uintptr_t)d | (uintptr_t)s) & 15) == 0) {
        // (the buffer row is not read again before the update phase: streamed stores)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        for (size_t i = tid; i < total / 16; i += nthr) __builtin_nontemporal_store(((const u32x4 *)s)[i], (u32x4 *)d + i);
    } else {
        for (size_t i = tid; i < total / 4; i += nthr) ((uint32_t *)d)[i] = ((const uint32_t *)s)[i];
    }
}

extern "C" int parc_record_step(void *stream, int n_envs, const int64_t *head, int n_fields, const parc_record_field_t *fields) {
    if (n_envs <= 0 || n_fields <= 0 || n_fields > PARC_RECORD_MAX_FIELDS || !head || !fields) return PARC_EINVAL;
    record_fields_t args;
    int max_row = 0;
    for (int i = 0; i < n_fields; ++i) {
        args.f[i] = fields[i];
        if (fields[i].row_bytes <= 0 || (fields[i].row_bytes & 3) || !fields[i].src || !fields[i].dst) return PARC_EINVAL;
        if (fields[i].row_bytes > max_row) max_row = fields[i].row_bytes;
    }
    for (int i = n_fields; i < PARC_RECORD_MAX_FIELDS; ++i) args.f[i] = fields[0];
    size_t units = ((size_t)n_envs * (size_t)max_row + 15) / 16;
    unsigned gx = (unsigned)((units + 255) / 256);
    if (gx > 2048u) gx = 2048u;                                          // grid-stride beyond that
    hipLaunchKernelGGL(record_step_kernel, dim3(gx, (unsigned)n_fields), dim3(256), 0, (hipStream_t)stream, n_envs, head, args);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// The random numbers of one rollout step in ONE launch: Philox4x32-10 (Salmon et al. 2011; the generator torch's device RNG is built
// on), keyed by the seed, counter = (thread, 0, step counter lo, hi).  n_uniform floats in [0, 1) (24 random bits, like torch's
// uniform_) - the env's pool: xy-target resample, restart sampling - and n_normal floats ~ N(0, 1) (Box-Muller on pairs) - the
// policy's action noise.  state[0] = step counter, advanced by the launch's last workgroup; state[1] = its ticket (zero between launches).
// Replaces two torch generator launches per step, which inside a replayed hipGraph also cost two fills of the generator's
// seed / offset cells per replay.  Being the FIRST launch of a rollout step it can also carry the step's tick: tick_cell (optional) <-
// (tick_cell + 1) % tick_mod by the launch's last workgroup - ExperienceBuffer.inc (experience_buffer.py:41-44) for a write row that lives
// on the device.  (157 workgroups take the ticket here; taking it in the record launch - 14 000 workgroups on one atomic - cost 200 us.)
// =============================================================================================
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void rng_step_kernel(uint64_t seed, uint64_t *state, float *__restrict__ uniform_out, int64_t n_uniform,
                                                       float *__restrict__ normal_out, int64_t n_normal, int64_t *tick_cell, int tick_mod) {
    const uint64_t step = state[0];
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one Philox block = 4 outputs
    const int64_t qu = (n_uniform + 3) / 4, qn = (n_normal + 3) / 4;
    if (q < qu + qn) {
        uint32_t r[4];
        philox4x32_10((uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r);
        if (q < qu) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * q + k < n_uniform) uniform_out[4 * q + k] = (float)(r[k] >> 8) * (1.0f / 16777216.0f);
        } else {
            const int64_t b = 4 * (q - qu);
            float z[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float u1 = ((float)(r[2 * k] >> 8) + 1.0f) * (1.0f / 16777216.0f);       // (0, 1]
                const float u2 = (float)(r[2 * k + 1] >> 8) * (1.0f / 16777216.0f);
                const float rad = sqrtf(-2.0f * __logf(u1));
                float sn, cs;
                __sincosf(6.283185307179586f * u2, &sn, &cs);
                z[2 * k] = rad * cs;
                z[2 * k + 1] = rad * sn;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (b + k < n_normal) normal_out[b + k] = z[k];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *ticket = reinterpret_cast<unsigned long long *>(state + 1);
        if (atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1ull) {
            state[0] = step + 1;
            *ticket = 0ull;
            if (tick_cell) tick_cell[0] = (tick_cell[0] + 1) % (int64_t)tick_mod;      // nothing of this launch reads it
        }
    }
}

extern "C" int parc_rng_step(void *stream, uint64_t seed, uint64_t *state, float *uniform_out, int64_t n_uniform, float *normal_out,
                             int64_t n_normal, int64_t *tick_cell, int tick_mod) {
    if (!state || n_uniform < 0 || n_normal < 0 || (n_uniform > 0 && !uniform_out) || (n_normal > 0 && !normal_out) || (tick_cell && tick_mod < 1))
        return PARC_EINVAL;
    const int64_t quads = (n_uniform + 3) / 4 + (n_normal + 3) / 4;
    if (quads == 0 && !tick_cell) return PARC_OK;
    hipLaunchKernelGGL(rng_step_kernel, dim3((unsigned)((quads + 255) / 256 > 0 ? (quads + 255) / 256 : 1)), dim3(256), 0, (hipStream_t)stream, seed,
                       state, uniform_out, n_uniform, normal_out, n_normal, tick_cell, tick_mod);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// K21 episodic return tracker: DMPPOReturnTracker.update (learning/dm_ppo_return_tracker.py:6-99) in one launch.
// Accumulate the K reward terms and the episode length per env, fold the envs that finished into the running means (weights by
// episode count, as the reference), clear them.  Reductions in a fixed order (deterministic).
// =============================================================================================
#define TRK_THREADS 256
#define TRK_MAX_K 12
#define TRK_SLOTS (TRK_MAX_K + 2)          // K term sums | summed episode length | finished count

__device__ __forceinline__ float lerp_torch(float a, float b, float w) { return w < 0.5f ? a + w * (b - a) : b - (b - a) * (1.0f - w); }

// Two launches: (1) one env per thread, ceil(N / 256) workgroups, each parks the sums of its finished envs in the workspace (folded in a
// fixed order: wave shuffle, then the four wave rows); (2) one small workgroup adds the workgroups' rows in workgroup order and updates
// the running means.  The result does not depend on scheduling.  (Round 1-3: one launch, the last workgroup to take a ticket did step
// 2 - the device-scope fences of that hand-over made it 13.6 us; one workgroup of 1024 threads over all envs - round 4, tried - 20-23 us:
// 180 KB through one CU.)
#undef TRK_THREADS
#define TRK_THREADS 256
__global__ __launch_bounds__(TRK_THREADS) void return_tracker_kernel(int n_envs, int K, const float *__restrict__ rewards, int64_t reward_stride,
                                                                     const int32_t *__restrict__ done, float *return_buf, int64_t *ep_len,
                                                                     int64_t *eps_per_env, float *__restrict__ part) {
    __shared__ float s_part[TRK_THREADS / 64][TRK_SLOTS];
    const int tid = threadIdx.x;
    const int e = blockIdx.x * TRK_THREADS + tid;
    float acc[TRK_SLOTS];
#pragma unroll
    for (int k = 0; k < TRK_SLOTS; ++k) acc[k] = 0.f;
    if (e < n_envs) {
        const bool fin = done[e] != 0;
        const int64_t len = ep_len[e] + 1;
        if (fin) {
            acc[TRK_MAX_K] = (float)len;
            acc[TRK_MAX_K + 1] = 1.0f;
            eps_per_env[e] += 1;
        }
        ep_len[e] = fin ? 0 : len;
#pragma unroll
        for (int k = 0; k < TRK_MAX_K; ++k) {
            if (k < K) {
                const float v = return_buf[(size_t)k * n_envs + e] + rewards[(size_t)k * reward_stride + e];
                if (fin) acc[k] = v;
                return_buf[(size_t)k * n_envs + e] = fin ? 0.f : v;
            }
        }
    }
    const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
    for (int k = 0; k < TRK_SLOTS; ++k) {
        const float v = wave_sum(acc[k]);
        if (ln == 0) s_part[wv][k] = v;
    }
    __syncthreads();
    if (tid < TRK_SLOTS) {
        float v = 0.f;
        for (int w = 0; w < TRK_THREADS / 64; ++w) v += s_part[w][tid];
        part[(size_t)blockIdx.x * TRK_SLOTS + tid] = v;
    }
}

__global__ __launch_bounds__(64) void return_tracker_fold_kernel(int groups, int K, const float *__restrict__ part, float *mean_return,
                                                                 float *mean_ep_len, double *episodes) {
    __shared__ float s_tot[TRK_SLOTS];
    const int tid = threadIdx.x;
    if (tid < TRK_SLOTS) {
        float v = 0.f;
        for (int g = 0; g < groups; ++g) v += part[(size_t)g * TRK_SLOTS + tid];
        s_tot[tid] = v;
    }
    __syncthreads();
    // DMPPOReturnTracker.update's running means (dm_ppo_return_tracker.py:60-99): weights by episode count
    const float n_new = s_tot[TRK_MAX_K + 1];
    if (n_new > 0.f) {
        const double new_count = episodes[0] + (double)n_new;
        const float w_new = (float)((double)n_new / (new_count < 1.0 ? 1.0 : new_count));
        if (tid < K) mean_return[tid] = lerp_torch(mean_return[tid], s_tot[tid] / n_new, w_new);
        if (tid == K) mean_ep_len[0] = lerp_torch(mean_ep_len[0], s_tot[TRK_MAX_K] / n_new, w_new);
        __syncthreads();
        if (tid == 0) episodes[0] = new_count;
    }
}

extern "C" int64_t parc_return_tracker_workspace_floats(int n_envs) {
    if (n_envs <= 0) return -1;
    return (int64_t)((n_envs + TRK_THREADS - 1) / TRK_THREADS) * TRK_SLOTS + 4;
}

extern "C" int parc_return_tracker_update(void *stream, int n_envs, int K, const float *rewards, int64_t reward_stride, const int32_t *done,
                                          float *return_buf, int64_t *ep_len, int64_t *eps_per_env, float *mean_return, float *mean_ep_len,
                                          double *episodes, float *workspace) {
    if (n_envs <= 0 || K <= 0 || K > TRK_MAX_K || reward_stride < n_envs || !workspace) return PARC_EINVAL;
    const int groups = (n_envs + TRK_THREADS - 1) / TRK_THREADS;
    hipLaunchKernelGGL(return_tracker_kernel, dim3(groups), dim3(TRK_THREADS), 0, (hipStream_t)stream, n_envs, K, rewards, reward_stride, done,
                       return_buf, ep_len, eps_per_env, workspace);
    hipLaunchKernelGGL(return_tracker_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, groups, K, workspace, mean_return, mean_ep_len, episodes);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// Gradient clipping by global norm (torch.nn.utils.clip_grad_norm_ as MPOptimizer applies it to its flat gradient buffer):
// x *= min(max_norm / (norm + 1e-6), 1) with norm read from device memory - one launch instead of add, reciprocal, mul, clamp, mul.
// =============================================================================================
__global__ __launch_bounds__(256) void scale_by_clipped_norm_kernel(size_t n, float *x, const float *__restrict__ norm, float max_norm) {
    const float s = fminf(max_norm / (norm[0] + 1e-6f), 1.0f);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= s;
}

extern "C" int parc_scale_by_clipped_norm(void *stream, int64_t n, float *x, const float *norm, float max_norm) {
    if (n < 0 || !x || !norm || !(max_norm > 0.f)) return PARC_EINVAL;
    if (n == 0) return PARC_OK;
    size_t blocks = ((size_t)n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_by_clipped_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (size_t)n, x, norm, max_norm);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// K20 optimizer step over flat buffers: MPOptimizer.step (learning/mp_optimizer.py:20-40) = clip_grad_norm_ + SGD with momentum, as
// TWO passes over the 10.6 M parameters instead of four (norm, scale, multi-tensor SGD, + a zero fill): pass 1 the squared norm of
// the flat gradient (1024 block partials), pass 2 every block first adds the partials in block order (fixed order, the same value
// in every block), coef = min(max_norm / (norm + 1e-6), 1), then for its elements g' = coef g (+ wd p), m = mu m + g', p -= lr m
// (torch.optim.SGD, dampening 0, no Nesterov; a zero momentum buffer reproduces its first-step rule buf = g').
// =============================================================================================
#define SGD_PARTS 1024
__global__ __launch_bounds__(256) void sumsq_partial_kernel(size_t n4, const float4 *__restrict__ g, size_t n, const float *__restrict__ gs,
                                                            float *__restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g[i];
        acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {       // tail elements beyond the last whole float4
        const float t = gs[4 * n4 + threadIdx.x];
        acc = fmaf(t, t, acc);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sgd_momentum_clip_kernel(size_t n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                                const float *__restrict__ partial, float max_norm, float lr, float mu, float wd,
                                                                float *norm_out) {
    __shared__ float s_red[256];
    float coef = 1.0f;
    if (max_norm > 0.f) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < SGD_PARTS / 256; ++k) a += partial[threadIdx.x * (SGD_PARTS / 256) + k];      // consecutive partials per thread
        s_red[threadIdx.x] = a;
        __syncthreads();
        for (int w = 128; w >= 1; w >>= 1) {          // pairwise tree over the 256 per-thread sums: the same order in every block
            if ((int)threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
            __syncthreads();
        }
        const float norm = sqrtf(s_red[0]);
        coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        float gi = coef * g[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = fmaf(mu, m[i], gi);
        m[i] = mi;
        p[i] = fmaf(-lr, mi, pi);
    }
}

extern "C" int64_t parc_sgd_workspace_floats(void) { return SGD_PARTS; }

extern "C" int parc_sgd_momentum_step(void *stream, int64_t n, float *params, const float *grad, float *momentum_buf, float max_norm, float lr,
                                      float momentum, float weight_decay, float *workspace, float *norm_out) {
    if (n < 0 || !params || !grad || !momentum_buf || !workspace || ((uintptr_t)grad & 15)) return PARC_EINVAL;
    if (n == 0) return PARC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (max_norm > 0.f)
        hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SGD_PARTS), dim3(256), 0, st, (size_t)n / 4, (const float4 *)grad, (size_t)n, grad, workspace);
    size_t blocks = ((size_t)n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(sgd_momentum_clip_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (size_t)n, params, grad, momentum_buf, workspace, max_norm,
                       lr, momentum, weight_decay, norm_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// K12 observation normalisation: Normalizer.normalize (learning/normalizer.py:60-63)  out = clamp((x - mean) / std, -clip, clip)
// in one pass (torch: subtract, divide, clamp = three passes).  Same fp32 operations, so the result is bit-identical.
// =============================================================================================
__global__ __launch_bounds__(256) void normalize_clamp_kernel(size_t n4, int dim4, const float4 *__restrict__ x, const float4 *__restrict__ mean,
                                                              const float4 *__restrict__ stdv, float clip, float4 *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (size_t)dim4);
        const float4 v = x[i], m = mean[c], s = stdv[c];
        float4 o;
        o.x = fminf(fmaxf((v.x - m.x) / s.x, -clip), clip);
        o.y = fminf(fmaxf((v.y - m.y) / s.y, -clip), clip);
        o.z = fminf(fmaxf((v.z - m.z) / s.z, -clip), clip);
        o.w = fminf(fmaxf((v.w - m.w) / s.w, -clip), clip);
        out[i] = o;
    }
}

extern "C" int parc_normalize_clamp(void *stream, int64_t rows, int dim, const float *x, const float *mean, const float *stdv, float clip,
                                    float *out) {
    if (rows < 0 || dim <= 0 || (dim & 3) || (((uintptr_t)x | (uintptr_t)mean | (uintptr_t)stdv | (uintptr_t)out) & 15)) return PARC_EINVAL;
    if (rows == 0) return PARC_OK;
    const size_t n4 = (size_t)rows * (size_t)(dim / 4);
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(normalize_clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n4, dim / 4, (const float4 *)x,
                       (const float4 *)mean, (const float4 *)stdv, clip, (float4 *)out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// K22 running-moment statistics: Normalizer.record (learning/normalizer.py:28-34)  acc[0] += sum_rows x, acc[1] += sum_rows x^2.
// torch: two column reductions + a square + two adds (and their temporaries) per call; here one pass over x.  Stage 1: a
// workgroup owns MOM_ROWS rows x 256 columns (64 float4 lanes x 4 row groups), sums in registers, folds the row groups through
// LDS and writes one partial row; stage 2 adds the partial rows of a column in chunk order into acc.  Fixed summation order:
// the result does not depend on scheduling.
// =============================================================================================
#define MOM_ROWS 64
__global__ __launch_bounds__(256) void moments_partial_kernel(int rows, int dim4, const float4 *__restrict__ x, float4 *__restrict__ partial) {
    __shared__ float4 red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * MOM_ROWS, r1 = min(r0 + MOM_ROWS, rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (c < dim4) {
        for (int r = r0 + rg; r < r1; r += 4) {
            const float4 v = x[(size_t)r * dim4 + c];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
        }
    }
    red[0][rg][lane] = s;
    red[1][rg][lane] = q;
    __syncthreads();
    if (rg < 2 && c < dim4) {       // wave 0 folds the sums, wave 1 the sums of squares
        const float4 a = red[rg][0][lane], b = red[rg][1][lane], d = red[rg][2][lane], e = red[rg][3][lane];
        float4 o;
        o.x = (a.x + b.x) + (d.x + e.x); o.y = (a.y + b.y) + (d.y + e.y); o.z = (a.z + b.z) + (d.z + e.z); o.w = (a.w + b.w) + (d.w + e.w);
        partial[((size_t)blockIdx.y * 2 + rg) * dim4 + c] = o;
    }
}

// The rollout step's three passes over the observation rows in ONE (parc_obs_ingest): Normalizer.normalize for the policy forward
// (normalizer.py:60-63; same fp32 operations as normalize_clamp_kernel: bit-identical), ExperienceBuffer.record of the raw rows
// (experience_buffer.py:55-59) into time row *copy_row, and stage 1 of Normalizer.record (the kernel above: same mapping, same
// summation order, hence the same partial rows).  The rows are read once instead of three times.
__global__ __launch_bounds__(256) void obs_ingest_kernel(int rows, int dim4, const float4 *__restrict__ x, const float4 *__restrict__ mean,
                                                         const float4 *__restrict__ stdv, float clip, float4 *__restrict__ norm_out,
                                                         float4 *copy_dst, const int64_t *__restrict__ copy_row, float4 *partial) {
    __shared__ float4 red[2][4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * MOM_ROWS, r1 = min(r0 + MOM_ROWS, rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (c < dim4) {
        const float4 m = mean[c], sd = stdv[c];
        float4 *dst = copy_dst ? copy_dst + (size_t)copy_row[0] * (size_t)rows * dim4 : nullptr;
        for (int r = r0 + rg; r < r1; r += 4) {
            const size_t i = (size_t)r * dim4 + c;
            const float4 v = x[i];
            float4 o;
            o.x = fminf(fmaxf((v.x - m.x) / sd.x, -clip), clip);
            o.y = fminf(fmaxf((v.y - m.y) / sd.y, -clip), clip);
            o.z = fminf(fmaxf((v.z - m.z) / sd.z, -clip), clip);
            o.w = fminf(fmaxf((v.w - m.w) / sd.w, -clip), clip);
            norm_out[i] = o;
            if (dst) {                       // (the buffer row is not read again before the update phase: streamed store)
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4 *>(dst) + i);
            }
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
        }
    }
    if (!partial) return;               // (uniform: a launch argument)
    red[0][rg][lane] = s;
    red[1][rg][lane] = q;
    __syncthreads();
    if (rg < 2 && c < dim4) {
        const float4 a = red[rg][0][lane], b = red[rg][1][lane], d = red[rg][2][lane], e = red[rg][3][lane];
        float4 o;
        o.x = (a.x + b.x) + (d.x + e.x); o.y = (a.y + b.y) + (d.y + e.y); o.z = (a.z + b.z) + (d.z + e.z); o.w = (a.w + b.w) + (d.w + e.w);
        partial[((size_t)blockIdx.y * 2 + rg) * dim4 + c] = o;
    }
}

// (stage 2 stays a launch of its own: folded into stage 1 behind per-column-block tickets - round 4 - the launch took 38 us instead of
// 6.8 + 4.9: the partial rows must be visible to the last workgroup, and a device-scope release fence per workgroup costs more than
// the launch it saves)
__global__ __launch_bounds__(256) void moments_final_kernel(int chunks, int dim4, const float4 *__restrict__ partial, float4 *acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over [2, dim4]
    if (i >= 2 * dim4) return;
    const int which = i / dim4, c = i - which * dim4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
    for (int k = 0; k < chunks; ++k) {          // (unrolled: the loads of 16 chunks are in flight together; the adds stay in chunk order)
        const float4 v = partial[((size_t)k * 2 + which) * dim4 + c];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float4 a = acc[i];
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    acc[i] = a;
}

extern "C" int64_t parc_moments_workspace_floats(int64_t rows, int dim) {
    if (rows < 0 || dim <= 0) return -1;
    return ((rows + MOM_ROWS - 1) / MOM_ROWS) * 2 * (int64_t)dim;
}

extern "C" int parc_obs_ingest(void *stream, int64_t rows, int dim, const float *x, const float *mean, const float *stdv, float clip, float *norm_out,
                               float *copy_dst, const int64_t *copy_row, float *acc, float *workspace) {
    if (rows < 0 || dim <= 0 || (dim & 3) || !x || !mean || !stdv || !norm_out || (copy_dst && !copy_row) || (acc && !workspace) ||
        (((uintptr_t)x | (uintptr_t)mean | (uintptr_t)stdv | (uintptr_t)norm_out | (uintptr_t)copy_dst | (uintptr_t)acc | (uintptr_t)workspace) & 15))
        return PARC_EINVAL;
    if (rows == 0) return PARC_OK;
    if (rows > (int64_t)MOM_ROWS * 65535) return PARC_EUNSUPPORTED;
    const int dim4 = dim / 4, chunks = (int)((rows + MOM_ROWS - 1) / MOM_ROWS);
    hipLaunchKernelGGL(obs_ingest_kernel, dim3((dim4 + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (int)rows, dim4, (const float4 *)x,
                       (const float4 *)mean, (const float4 *)stdv, clip, (float4 *)norm_out, (float4 *)copy_dst, copy_row,
                       acc ? (float4 *)workspace : (float4 *)nullptr);
    if (acc)
        hipLaunchKernelGGL(moments_final_kernel, dim3((2 * dim4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, chunks, dim4,
                           (const float4 *)workspace, (float4 *)acc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

extern "C" int parc_moments_accumulate(void *stream, int64_t rows, int dim, const float *x, float *acc, float *workspace) {
    if (rows < 0 || dim <= 0 || (dim & 3) || (((uintptr_t)x | (uintptr_t)acc | (uintptr_t)workspace) & 15)) return PARC_EINVAL;
    if (rows == 0) return PARC_OK;
    if (rows > (int64_t)MOM_ROWS * 65535) return PARC_EUNSUPPORTED;
    const int dim4 = dim / 4, chunks = (int)((rows + MOM_ROWS - 1) / MOM_ROWS);
    hipLaunchKernelGGL(moments_partial_kernel, dim3((dim4 + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (int)rows, dim4, (const float4 *)x,
                       (float4 *)workspace);
    hipLaunchKernelGGL(moments_final_kernel, dim3((2 * dim4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, chunks, dim4,
                       (const float4 *)workspace, (float4 *)acc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// Backward of y = relu(x W^T + b) between the two GEMMs: g = gy * (y > 0), written over gy, and the bias gradient db = column sums
// of g, in ONE pass over the [rows, dim] activations (autograd runs threshold_backward, then a separate reduction that re-reads g,
// then an accumulate into .grad).  256 columns x RB_ROWS rows per workgroup, float4 lanes, the 4 waves take rows round-robin and fold
// their column sums through LDS; stage 2 adds the per-chunk partial rows in chunk order (fixed summation order, no atomics) and
// OVERWRITES db.  The forward pass is learning/nets/fc_3layers_2048units.py:4-22 of the reference; this is its derivative.
// =============================================================================================
#define RB_ROWS 128
__global__ __launch_bounds__(256) void relu_bwd_bias_partial_kernel(int rows, int dim4, float4 *__restrict__ gy, const float4 *__restrict__ y,
                                                                    float4 *__restrict__ partial) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * RB_ROWS, r1 = min(r0 + RB_ROWS, rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < dim4) {
#pragma unroll 4
        for (int r = r0 + rg; r < r1; r += 4) {
            const size_t i = (size_t)r * dim4 + c;
            float4 g = gy[i];
            const float4 a = y[i];
            g.x = a.x > 0.f ? g.x : 0.f; g.y = a.y > 0.f ? g.y : 0.f; g.z = a.z > 0.f ? g.z : 0.f; g.w = a.w > 0.f ? g.w : 0.f;
            gy[i] = g;
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        }
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && c < dim4) {
        const float4 a = red[0][lane], b = red[1][lane], d = red[2][lane], e = red[3][lane];
        float4 o;
        o.x = (a.x + b.x) + (d.x + e.x); o.y = (a.y + b.y) + (d.y + e.y); o.z = (a.z + b.z) + (d.z + e.z); o.w = (a.w + b.w) + (d.w + e.w);
        partial[(size_t)blockIdx.y * dim4 + c] = o;
    }
}

__global__ __launch_bounds__(256) void colsum_final_kernel(int chunks, int dim, const float *__restrict__ partial, float *__restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dim) return;
    float t = 0.f;
#pragma unroll 16
    for (int k = 0; k < chunks; ++k) t += partial[(size_t)k * dim + c];
    out[c] = t;
}

// out[c] = sum_r w[r] * x[r, c]: the weight gradient of a Linear layer with ONE output (the value head: dW = g_pred^T h), which as
// a GEMM with M = 1 costs the library 62 us and as its transposed gemv 311 us for 33 MB of reading; same two stages as above.
__global__ __launch_bounds__(256) void weighted_colsum_partial_kernel(int rows, int dim4, const float4 *__restrict__ x, const float *__restrict__ w,
                                                                      float4 *__restrict__ partial) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * RB_ROWS, r1 = min(r0 + RB_ROWS, rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < dim4) {
#pragma unroll 4
        for (int r = r0 + rg; r < r1; r += 4) {
            const float4 v = x[(size_t)r * dim4 + c];
            const float wr = w[r];
            s.x = fmaf(wr, v.x, s.x); s.y = fmaf(wr, v.y, s.y); s.z = fmaf(wr, v.z, s.z); s.w = fmaf(wr, v.w, s.w);
        }
    }
    red[rg][lane] = s;
    __syncthreads();
    if (rg == 0 && c < dim4) {
        const float4 a = red[0][lane], b = red[1][lane], d = red[2][lane], e = red[3][lane];
        float4 o;
        o.x = (a.x + b.x) + (d.x + e.x); o.y = (a.y + b.y) + (d.y + e.y); o.z = (a.z + b.z) + (d.z + e.z); o.w = (a.w + b.w) + (d.w + e.w);
        partial[(size_t)blockIdx.y * dim4 + c] = o;
    }
}

extern "C" int parc_weighted_colsum(void *stream, int64_t rows, int dim, const float *x, const float *w, float *out, float *workspace) {
    if (rows < 0 || dim <= 0 || (dim & 3) || !x || !w || !out || !workspace || (((uintptr_t)x | (uintptr_t)workspace) & 15)) return PARC_EINVAL;
    if (rows > (int64_t)RB_ROWS * 65535) return PARC_EUNSUPPORTED;
    const int dim4 = dim / 4, chunks = (int)((rows + RB_ROWS - 1) / RB_ROWS);
    if (rows > 0)
        hipLaunchKernelGGL(weighted_colsum_partial_kernel, dim3((dim4 + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (int)rows, dim4,
                           (const float4 *)x, w, (float4 *)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, chunks, dim, workspace, out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

extern "C" int64_t parc_relu_bwd_workspace_floats(int64_t rows, int dim) {
    if (rows < 0 || dim <= 0) return -1;
    return ((rows + RB_ROWS - 1) / RB_ROWS) * (int64_t)dim;
}

extern "C" int parc_relu_bwd_bias_grad(void *stream, int64_t rows, int dim, float *gy, const float *y, float *db, float *workspace) {
    if (rows < 0 || dim <= 0 || (dim & 3) || !gy || !y || !db || !workspace || (((uintptr_t)gy | (uintptr_t)y | (uintptr_t)workspace) & 15))
        return PARC_EINVAL;
    if (rows > (int64_t)RB_ROWS * 65535) return PARC_EUNSUPPORTED;
    const int dim4 = dim / 4, chunks = (int)((rows + RB_ROWS - 1) / RB_ROWS);
    if (rows > 0)
        hipLaunchKernelGGL(relu_bwd_bias_partial_kernel, dim3((dim4 + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, (int)rows, dim4,
                           (float4 *)gy, (const float4 *)y, (float4 *)workspace);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((dim + 255) / 256), dim3(256), 0, (hipStream_t)stream, chunks, dim, workspace, db);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

// =============================================================================================
// K14 action head of the rollout: PPOAgent._decide_action (learning/ppo_agent.py:87-119) after the actor MLP.
// norm_a = mean + std * noise where the env explores (mask 1), the mode otherwise; a_logp = log N(norm_a; mean, std);
// action = a_mean + a_std * norm_a (Normalizer.unnormalize).  One thread per env.
// =============================================================================================
__global__ __launch_bounds__(256) void action_head_kernel(int n, int A, const float *__restrict__ mean, const float *__restrict__ logstd,
                                                          const float *__restrict__ noise, const float *__restrict__ explore,
                                                          const float *__restrict__ a_mean, const float *__restrict__ a_std, float *action,
                                                          float *logp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool ex = explore[i] == 1.0f;
    float acc = 0.f, sum_ls = 0.f;
    for (int j = 0; j < A; ++j) {
        const float ls = logstd[j], sd = __expf(ls);
        const float mu = mean[(size_t)i * A + j];
        const float na = ex ? mu + sd * noise[(size_t)i * A + j] : mu;
        const float z = (na - mu) / sd;
        acc = fmaf(z, z, acc);
        sum_ls += ls;
        action[(size_t)i * A + j] = na * a_std[j] + a_mean[j];
    }
    logp[i] = -0.5f * acc + (-0.5f * (float)A * 1.8378770664093453f - sum_ls);
}

// A <= 32 (the humanoid has 28 actuated dofs): 32 lanes per env, lane j = action dimension j - coalesced reads of the [n, A] rows and
// 512 workgroups at 4096 envs instead of 16; the two sums over j are 5-step xor-shuffle reductions inside the 32-lane group.
// (rec.head != NULL: the same launch also writes what ExperienceBuffer.record stores of this moment - action, a_logp,
// rand_action_mask and the contact forces the env holds before the step - into time row *head of their [T, n, ...] buffers:
// base_agent's _record_data_pre_step (ppo_agent.py:121-125, dm_ppo_agent.py:289-299) without a launch of its own)
struct action_record_t {
    float *action, *logp, *mask, *forces;      // [T, n, A] [T, n] [T, n] [T, n, n_forces]
    const float *forces_src;                    // [n, n_forces]
    int n_forces;
    const int64_t *head;
};

__global__ __launch_bounds__(256) void action_head32_kernel(int n, int A, const float *__restrict__ mean, const float *__restrict__ logstd,
                                                            const float *__restrict__ noise, const float *__restrict__ explore,
                                                            const float *__restrict__ a_mean, const float *__restrict__ a_std, float *action,
                                                            float *logp, action_record_t rec) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = min(gid >> 5, n - 1), j = gid & 31;           // (tail groups redo the last env; they skip the stores)
    const bool live = (gid >> 5) < n, valid = j < A;
    float zz = 0.f, ls = 0.f;
    if (valid) {
        ls = logstd[j];
        const float sd = __expf(ls);
        const float mu = mean[(size_t)i * A + j];
        const float na = explore[i] == 1.0f ? mu + sd * noise[(size_t)i * A + j] : mu;
        const float z = (na - mu) / sd;
        zz = z * z;
        if (live) {
            const float a = na * a_std[j] + a_mean[j];
            action[(size_t)i * A + j] = a;
            if (rec.head) rec.action[((size_t)rec.head[0] * n + i) * A + j] = a;
        }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
        zz += __shfl_xor(zz, o, 32);
        ls += __shfl_xor(ls, o, 32);
    }
    if (live && j == 0) {
        const float lp = -0.5f * zz + (-0.5f * (float)A * 1.8378770664093453f - ls);
        logp[i] = lp;
        if (rec.head) {
            const size_t row = (size_t)rec.head[0] * n + i;
            rec.logp[row] = lp;
            rec.mask[row] = explore[i];
        }
    }
    if (live && rec.head)
        for (int k = j; k < rec.n_forces; k += 32) rec.forces[((size_t)rec.head[0] * n + i) * rec.n_forces + k] = rec.forces_src[(size_t)i * rec.n_forces + k];
}

extern "C" int parc_action_head(void *stream, int n, int A, const float *mean, const float *logstd, const float *noise, const float *explore,
                                const float *a_mean, const float *a_std, float *action, float *logp) {
    if (n <= 0 || A <= 0) return PARC_EINVAL;
    if (A <= 32) {
        const long long threads = (long long)n * 32;
        action_record_t none = {};
        hipLaunchKernelGGL(action_head32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, A, mean, logstd, noise,
                           explore, a_mean, a_std, action, logp, none);
        hipError_t e32 = hipGetLastError();
        return e32 == hipSuccess ? PARC_OK : (int)e32;
    }
    hipLaunchKernelGGL(action_head_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, A, mean, logstd, noise, explore, a_mean,
                       a_std, action, logp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

extern "C" int parc_action_head_record(void *stream, int n, int A, const float *mean, const float *logstd, const float *noise, const float *explore,
                                       const float *a_mean, const float *a_std, float *action, float *logp, float *rec_action, float *rec_logp,
                                       float *rec_mask, const float *forces_src, float *rec_forces, int n_forces, const int64_t *head) {
    if (n <= 0 || A <= 0 || !rec_action || !rec_logp || !rec_mask || !forces_src || !rec_forces || n_forces < 0 || !head) return PARC_EINVAL;
    if (A > 32) return PARC_EUNSUPPORTED;
    action_record_t rec = {rec_action, rec_logp, rec_mask, rec_forces, forces_src, n_forces, head};
    const long long threads = (long long)n * 32;
    hipLaunchKernelGGL(action_head32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, A, mean, logstd, noise, explore,
                       a_mean, a_std, action, logp, rec);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}
