/* Extra entry points of the DIAGNOSTICS build of the library (libparc_hip_diag.so, built by tools/parc_diag.py with -DPARC_DIAG_BUILD).
 * None of this is in the product library (parc_amd/lib/libparc_hip.so) or in include/: measurement only. */
#ifndef PARC_DIAG_H
#define PARC_DIAG_H
#include "../include/parc_hip.h"
#include "../include/parc_sim.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Standalone heightmap kernel (parc_refresh_obs_hfs), process-global:
 * envs per workgroup (1|2|4|8); 128-thread env groups per workgroup (1|2|4|8, two envs per group); timing-only ablations 0..5
 * (outputs wrong for != 0). */
int parc_tune_hf_envs_per_block(int envs_per_block);
int parc_tune_hf_groups(int groups);
int parc_tune_hf_ablation(int variant);

/* occupancy probe of track_post_kernel: bytes of dynamic LDS added to every workgroup (process-global; 0 = the product's launch) */
int parc_tune_post_lds_pad(int bytes);

/* parc_track_post_step of this build also honours these bits of `what` (timing only: the launch's outputs are garbage) */
#define PARC_DIAG_POST_NO_TARGET_WAVES 0x10000
#define PARC_DIAG_POST_NO_REFERENCE_WAVE 0x20000
#define PARC_DIAG_POST_NO_CHARACTER_WAVE 0x40000
#define PARC_DIAG_POST_NO_HEIGHTMAP_WAVE 0x80000
#define PARC_DIAG_POST_RETURN_AT_ENTRY 0x100000
#define PARC_DIAG_POST_RETURN_BEFORE_BARRIER 0x200000
#define PARC_DIAG_POST_TARGET_UP_TO_SLERP 0x400000
#define PARC_DIAG_POST_TARGET_UP_TO_TREE_WALK 0x800000
#define PARC_DIAG_POST_TARGET_NO_STORES 0x1000000

/* The simulator step (same contract as parc_sim_step, include/parc_sim.h) on the one-env-per-lane reference formulation
 * (parc_sim_core.h, the source oracle/sim_host.cpp builds for the host); threads = envs per workgroup (8|16|32|64). */
int parc_diag_sim_step_env_per_lane(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                                    float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets,
                                    const float *action, const float *action_low, const float *action_high, int n_substeps, float h,
                                    int threads);

#ifdef __cplusplus
}
#endif
#endif
