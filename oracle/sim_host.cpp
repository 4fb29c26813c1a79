// HOST build of the simulator core (parc_amd/csrc/parc_sim_core.h) -- TEST INFRASTRUCTURE ONLY.
// The dynamics have no reference arithmetic (Isaac Gym is an absent binary), so this is not an oracle of the
// reference: it is the same source compiled for the CPU, used by tests to check physical invariants without a
// GPU, to run sanitizers, and to cross-check the device build.  Same signatures as include/parc_sim.h but all
// pointers are HOST pointers.
#include <string.h>

#include "../parc_amd/csrc/parc_sim_core.h"

// >= 0: the per-env work arrays start from this byte pattern (0xFF = NaN everywhere) instead of whatever the stack holds.  A
// read of an element that the algorithm never wrote then shows up as a changed (NaN) result: tests compare fill 0x00 with 0xFF.
static int g_fill = -1;
extern "C" void sim_host_set_fill(int byte) { g_fill = byte; }

extern "C" int sim_host_step(const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state, float *dof_state,
                             float *rigid_body_state, float *contact_forces, const float *env_offsets, const float *action,
                             const float *action_low, const float *action_high, int n_substeps, float h) {
    const int B = model->num_bodies, D = model->dof_size;
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_envs; ++e) {
        parc_sim::Scratch s;
        if (g_fill >= 0) memset((void *)&s, g_fill, sizeof s);
        parc_sim::env_step(*model, terrain, env_offsets + 3 * (size_t)e, root_state + 13 * (size_t)e, dof_state + 2 * (size_t)D * e,
                           rigid_body_state + 13 * (size_t)B * e, contact_forces + 3 * (size_t)B * e, action + (size_t)D * e, action_low,
                           action_high, n_substeps, h, s);
    }
    return 0;
}

extern "C" int sim_host_refresh_bodies(const parc_sim_model_t *model, int n_envs, const float *root_state, const float *dof_state,
                                       float *rigid_body_state, float *contact_forces) {
    const int B = model->num_bodies, D = model->dof_size;
    float zero[PARC_SIM_MAX_DOFS] = {0}, lo[PARC_SIM_MAX_DOFS], hi[PARC_SIM_MAX_DOFS];
    for (int d = 0; d < PARC_SIM_MAX_DOFS; ++d) { lo[d] = -1.f; hi[d] = 1.f; }
    for (int e = 0; e < n_envs; ++e) {
        parc_sim::State x;
        parc_sim::load_state(*model, root_state + 13 * (size_t)e, dof_state + 2 * (size_t)D * e, zero, lo, hi, x);
        parc_sim::publish_bodies(*model, x, rigid_body_state + 13 * (size_t)B * e, contact_forces + 3 * (size_t)B * e);
    }
    return 0;
}

// Penetration depth of every collision sample sphere at a given kinematic state (0 = no contact): the geometric half of the
// contact model alone, used by the kinematic-replay test (a reference clip replayed on its own terrain must not sink into it).
extern "C" int sim_host_penetration(const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, const float *root_state,
                                    const float *dof_state, const float *env_offsets, float *depth_out) {
    const int D = model->dof_size, S = model->num_spheres;
    float zero[PARC_SIM_MAX_DOFS] = {0}, lo[PARC_SIM_MAX_DOFS], hi[PARC_SIM_MAX_DOFS];
    for (int d = 0; d < PARC_SIM_MAX_DOFS; ++d) { lo[d] = -1.f; hi[d] = 1.f; }
    for (int e = 0; e < n_envs; ++e) {
        parc_sim::State x;
        parc_sim::Scratch s;
        parc_sim::load_state(*model, root_state + 13 * (size_t)e, dof_state + 2 * (size_t)D * e, zero, lo, hi, x);
        parc_sim::V3 off = parc_sim::ld(env_offsets + 3 * (size_t)e);
        parc_sim::pass1(*model, terrain, off, x, s, 1.0f / 120.0f);
        for (int k = 0; k < S; ++k) {
            const int b = model->sph_body[k];
            parc_sim::V3 pw = s.P[b] + parc_sim::mul(s.R[b], parc_sim::ld(model->sph_pos[k]));
            float depth;
            parc_sim::V3 n;
            depth_out[(size_t)e * S + k] = parc_sim::sphere_vs_columns(terrain, pw + off, model->sph_radius[k], depth, n) ? depth : 0.f;
        }
    }
    return 0;
}
