// Device entry points of the articulated-body simulator.  The product kernels are the body-per-lane ones below (16 lanes per env, 4 envs
// per 64-thread workgroup, parc_sim_bpl.h); the one-env-per-lane reference formulation is a separate translation unit
// (parc_sim_ref.hip) that is compiled into the diagnostics library only (tools/parc_diag.py, parc_diag_sim_step_env_per_lane).
#include <hip/hip_runtime.h>

#include "parc_sim_bpl.h"
#include "parc_sim_core.h"
#include "../../include/parc_sim.h"

// body-per-lane step: 16 lanes per env, 4 envs per 64-thread workgroup (parc_sim_bpl.h)
__global__ __launch_bounds__(64) void sim_step_bpl_kernel(const parc_sim_model_t *__restrict__ model, parc_terrain_t ter, int n_envs,
                                                          float *root_state, float *dof_state, float *rigid_body_state,
                                                          float *contact_forces, const float *__restrict__ env_offsets,
                                                          const float *__restrict__ action, const float *__restrict__ act_lo,
                                                          const float *__restrict__ act_hi, int n_sub, float h, int32_t *timestep,
                                                          float *time_buf, float step_dt) {
    using namespace parc_sim_bpl;
    __shared__ float lds[BPL_EPB][BPL_G * BPL_CONTRIB];
    __shared__ float ccache[64][BPL_CC_SLOTS * BPL_CC_FLOATS + 1];     // +1: odd row stride against bank conflicts
    const int g = threadIdx.x / BPL_G, b = threadIdx.x % BPL_G;
    const int e = min((int)blockIdx.x * BPL_EPB + g, n_envs - 1);      // tail groups recompute the last env (same values)
    // the model (4.9 KB of per-body / per-dof / per-sphere constants, read ~70 times per lane and substep) staged in LDS once per
    // workgroup: 100.9 -> 97.2 us per 4096-env step (profiles/r04_sim_step_variants.txt)
    __shared__ parc_sim_model_t s_model;
    {
        static_assert(sizeof(parc_sim_model_t) % 4 == 0, "copied as 32-bit words");
        const uint32_t *src = reinterpret_cast<const uint32_t *>(model);
        uint32_t *dst = reinterpret_cast<uint32_t *>(&s_model);
        for (unsigned i = threadIdx.x; i < sizeof(parc_sim_model_t) / 4; i += 64) dst[i] = src[i];
        __syncthreads();
    }
    const parc_sim_model_t &m = s_model;
    const int B = m.num_bodies, D = m.dof_size;
    step_lane(m, ter, b, root_state + 13 * (size_t)e, dof_state + 2 * (size_t)D * e, rigid_body_state + 13 * (size_t)B * e,
              contact_forces + 3 * (size_t)B * e, env_offsets + 3 * (size_t)e, action + (size_t)D * e, act_lo, act_hi, n_sub, h, lds[g],
              ccache[threadIdx.x]);
    // IGEnv._update_time (ig_env.py:862-865) for callers that ask for it: the env's step counter and clock advance with the simulator
    if (timestep && b == 0 && (int)blockIdx.x * BPL_EPB + g < n_envs) {
        const int ts = timestep[e] + 1;
        timestep[e] = ts;
        time_buf[e] = (float)ts * step_dt;
    }
}

// body-per-lane refresh: body poses / velocities from the state rows, for a list of envs (env_ids, n = list length), for
// the envs whose mask is set (device-side reset, n = env count), or for all (both null)
__global__ __launch_bounds__(64) void sim_refresh_bpl_kernel(const parc_sim_model_t *__restrict__ model, int n, const int64_t *__restrict__ env_ids,
                                                             const int32_t *__restrict__ mask, const float *__restrict__ root_state,
                                                             const float *__restrict__ dof_state, float *rigid_body_state, float *contact_forces) {
    using namespace parc_sim_bpl;
    const int g = threadIdx.x / BPL_G, b = threadIdx.x % BPL_G;
    const int k0 = (int)blockIdx.x * BPL_EPB;
    if (mask) {
        int any = 0;
#pragma unroll
        for (int k = 0; k < BPL_EPB; ++k)
            if (k0 + k < n) any |= mask[k0 + k];
        if (!any) return;                               // uniform: nothing flagged in this workgroup
    }
    const int k = min(k0 + g, n - 1);
    const int e = env_ids ? (int)env_ids[k] : k;
    const bool live = k0 + g < n && (!mask || mask[e] != 0);
    const parc_sim_model_t &m = *model;
    const int B = m.num_bodies, D = m.dof_size;
    const Lane L = load_lane(m, b);
    int maxd = L.depth;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) maxd = max(maxd, __shfl_xor(maxd, o, BPL_G));
    LState x;
    // (PD targets are irrelevant here: the action / bound pointers are only read, point them at the dof row)
    const float *drow = dof_state + 2 * (size_t)D * e;
    load_lane_state(m, L, b, root_state + 13 * (size_t)e, drow, drow, drow, drow, x);
    if (!live) return;        // whole 16-lane groups leave together; the sweeps below only shuffle inside a group
    store_lane_state<false>(L, b, maxd, x, nullptr, nullptr, rigid_body_state + 13 * (size_t)B * e, contact_forces + 3 * (size_t)B * e);
}

static_assert(PARC_SIM_MAX_BODIES <= BPL_G, "one body per lane: a 16-lane group holds one env");

static int sim_step_impl(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                         float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets, const float *action,
                         const float *action_low, const float *action_high, int n_substeps, float h, int32_t *timestep, float *time_buf,
                         float step_dt) {
    if (!model || n_envs < 0 || n_substeps <= 0 || !(h > 0.f) || !terrain.hf) return PARC_EINVAL;
    if (n_envs == 0) return PARC_OK;
    hipLaunchKernelGGL(sim_step_bpl_kernel, dim3((n_envs + BPL_EPB - 1) / BPL_EPB), dim3(64), 0, (hipStream_t)stream, model, terrain, n_envs,
                       root_state, dof_state, rigid_body_state, contact_forces, env_offsets, action, action_low, action_high, n_substeps, h,
                       timestep, time_buf, step_dt);
    hipError_t e1 = hipGetLastError();
    return e1 == hipSuccess ? PARC_OK : (int)e1;
}

extern "C" int parc_sim_step(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                             float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets,
                             const float *action, const float *action_low, const float *action_high, int n_substeps, float h) {
    return sim_step_impl(stream, model, terrain, n_envs, root_state, dof_state, rigid_body_state, contact_forces, env_offsets, action, action_low,
                         action_high, n_substeps, h, nullptr, nullptr, 0.f);
}

extern "C" int parc_sim_step_tick(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                                  float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets,
                                  const float *action, const float *action_low, const float *action_high, int n_substeps, float h,
                                  int32_t *timestep_buf, float *time_buf, float step_dt) {
    if (!timestep_buf || !time_buf) return PARC_EINVAL;
    return sim_step_impl(stream, model, terrain, n_envs, root_state, dof_state, rigid_body_state, contact_forces, env_offsets, action, action_low,
                         action_high, n_substeps, h, timestep_buf, time_buf, step_dt);
}

extern "C" int parc_sim_refresh_bodies(void *stream, const parc_sim_model_t *model, int n_envs, const int64_t *env_ids, int n_sel,
                                       const float *root_state, const float *dof_state, float *rigid_body_state, float *contact_forces) {
    if (!model || n_envs < 0) return PARC_EINVAL;
    int n = env_ids ? n_sel : n_envs;
    if (n <= 0) return n == 0 ? PARC_OK : PARC_EINVAL;
    hipLaunchKernelGGL(sim_refresh_bpl_kernel, dim3((n + BPL_EPB - 1) / BPL_EPB), dim3(64), 0, (hipStream_t)stream, model, n, env_ids,
                       (const int32_t *)nullptr, root_state, dof_state, rigid_body_state, contact_forces);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

extern "C" int parc_sim_refresh_bodies_masked(void *stream, const parc_sim_model_t *model, int n_envs, const int32_t *mask,
                                              const float *root_state, const float *dof_state, float *rigid_body_state,
                                              float *contact_forces) {
    if (!model || n_envs < 0 || !mask) return PARC_EINVAL;
    if (n_envs == 0) return PARC_OK;
    hipLaunchKernelGGL(sim_refresh_bpl_kernel, dim3((n_envs + BPL_EPB - 1) / BPL_EPB), dim3(64), 0, (hipStream_t)stream, model, n_envs,
                       (const int64_t *)nullptr, mask, root_state, dof_state, rigid_body_state, contact_forces);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}

extern "C" int parc_sim_abi(void) { return 1; }
