// The one-env-per-lane formulation of the simulator as a device kernel: the single-source reference core (parc_sim_core.h, also built
// for the host by oracle/sim_host.cpp) with 64 envs per workgroup and its per-body arrays in scratch.  NOT what the product launches
// (that is sim_step_bpl_kernel in parc_sim.hip) and NOT in the product library: this file is compiled into the diagnostics library only
// (tools/parc_diag.py, libparc_hip_diag.so; entry point parc_diag_sim_step_env_per_lane, tools/parc_diag.h), kept because every
// invariant test runs on both formulations and the device build of this one is compared with its host build.  It lives in its own translation unit because hipcc (ROCm 7.2, gfx950) miscompiles THIS
// kernel at -O3 (GVN scalar PRE on the fully unrolled 3x3 helpers, profiles/r02_sim_o3_bisect.txt): this file is built at -O2, the
// product kernels are not held back by it.
#include <hip/hip_runtime.h>

#include "parc_sim_core.h"
#include "../../include/parc_sim.h"

#define SIM_THREADS 64

__global__ __launch_bounds__(SIM_THREADS) void sim_step_kernel(const parc_sim_model_t *__restrict__ model, parc_terrain_t ter, int n_envs,
                                                               float *root_state, float *dof_state, float *rigid_body_state,
                                                               float *contact_forces, const float *__restrict__ env_offsets,
                                                               const float *__restrict__ action, const float *__restrict__ act_lo,
                                                               const float *__restrict__ act_hi, int n_sub, float h) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    const parc_sim_model_t &m = *model;
    const int B = m.num_bodies, D = m.dof_size;
    parc_sim::Scratch s;
    parc_sim::env_step(m, ter, env_offsets + 3 * (size_t)e, root_state + 13 * (size_t)e, dof_state + 2 * (size_t)D * e,
                       rigid_body_state + 13 * (size_t)B * e, contact_forces + 3 * (size_t)B * e, action + (size_t)D * e, act_lo, act_hi,
                       n_sub, h, s);
}

// threads = envs (= lanes) per workgroup (8, 16, 32 or 64): 4096 envs are only 64 full waves on a 1024-SIMD chip, so partially filled
// waves on more CUs can win.  A per-call argument: no process-global state.
extern "C" int parc_diag_sim_step_env_per_lane(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                                               float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets,
                                               const float *action, const float *action_low, const float *action_high, int n_substeps, float h,
                                               int threads) {
    if (!model || n_envs < 0 || n_substeps <= 0 || !(h > 0.f) || !terrain.hf) return PARC_EINVAL;
    if (threads != 8 && threads != 16 && threads != 32 && threads != 64) return PARC_EINVAL;
    if (n_envs == 0) return PARC_OK;
    hipLaunchKernelGGL(sim_step_kernel, dim3((n_envs + threads - 1) / threads), dim3(threads), 0, (hipStream_t)stream, model, terrain, n_envs,
                       root_state, dof_state, rigid_body_state, contact_forces, env_offsets, action, action_low, action_high, n_substeps, h);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PARC_OK : (int)e;
}
