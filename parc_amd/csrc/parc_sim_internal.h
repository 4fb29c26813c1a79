// Internal link between the simulator's two translation units (not part of the C ABI, hidden from the library's exports).
#pragma once
#include "../../include/parc_sim.h"

__attribute__((visibility("hidden"))) int parc_sim_launch_env_per_lane(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain,
                                                                       int n_envs, float *root_state, float *dof_state, float *rigid_body_state,
                                                                       float *contact_forces, const float *env_offsets, const float *action,
                                                                       const float *action_low, const float *action_high, int n_substeps, float h);
