/*
 * parc_sim.h -- C-ABI of the articulated-body simulator inside libparc_hip.so.
 *
 * Stands in for the Isaac Gym calls of the reference's env step (paths relative to the reference root):
 *   gym.set_dof_position_target_tensor      envs/ig_char_env.py:489-495   (PD targets = clipped action)
 *   gym.simulate x sim_steps                envs/ig_env.py:830-837        (substeps from the YAML `sim:` block)
 *   gym.refresh_*_tensor                    envs/ig_env.py:850-860        (state published in the same tensors)
 * The state tensors ARE the simulator state (as with Isaac Gym): writing root_state / dof_state rows and
 * stepping is all a reset needs (envs/ig_env.py:693-721).
 * Dynamics parity is unpinned (no arithmetic reference exists outside the Isaac Gym binary); see DESIGN.md.
 */
#ifndef PARC_SIM_H
#define PARC_SIM_H

#include "parc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PARC_SIM_MAX_BODIES 16
#define PARC_SIM_MAX_DOFS 64
#define PARC_SIM_MAX_SPHERES 64

/* Character dynamics model, built on the host from the MJCF (data/assets/humanoid.xml): masses from geom
 * densities, PD gains = joint stiffness/damping, armature, limits (radians), motor gears as torque limits. */
typedef struct {
    int32_t num_bodies, dof_size, num_spheres, _pad;
    int32_t parent[PARC_SIM_MAX_BODIES];
    int32_t joint_type[PARC_SIM_MAX_BODIES];
    int32_t dof_idx[PARC_SIM_MAX_BODIES];
    float local_translation[PARC_SIM_MAX_BODIES][3];
    float local_rotation[PARC_SIM_MAX_BODIES][4];
    float joint_axis[PARC_SIM_MAX_BODIES][3];
    float mass[PARC_SIM_MAX_BODIES];
    float com[PARC_SIM_MAX_BODIES][3];          /* body frame */
    float inertia_o[PARC_SIM_MAX_BODIES][6];    /* about the body ORIGIN: xx xy xz yy yz zz */
    float kp[PARC_SIM_MAX_DOFS], kd[PARC_SIM_MAX_DOFS], armature[PARC_SIM_MAX_DOFS];
    float limit_lo[PARC_SIM_MAX_DOFS], limit_hi[PARC_SIM_MAX_DOFS], effort[PARC_SIM_MAX_DOFS];
    int32_t sph_body[PARC_SIM_MAX_SPHERES];      /* collision sample spheres */
    float sph_pos[PARC_SIM_MAX_SPHERES][3];
    float sph_radius[PARC_SIM_MAX_SPHERES];
    float gravity;                               /* 9.81 */
    float contact_kn, contact_cn, contact_ct, friction_mu, contact_max_pen;
    float limit_kp, limit_kd, max_angular_velocity;
    float angular_damping;                       /* 1/s, per link (asset_options.angular_damping, envs/ig_char_env.py:141): a couple
                                                    -c * I_com * omega on every body, i.e. d(omega)/dt = -c * omega for a free body */
    /* self-collision (Isaac Gym creates the actor with collision filter 0, envs/ig_char_env.py:105-113: links of one character
     * collide with each other except across a joint): one capsule per body (segment cap_p0..cap_p1 in the body frame, radius
     * cap_radius; <= 0 = none) and, per body, the set of bodies it is tested against (bit j of self_mask[b]) */
    float cap_p0[PARC_SIM_MAX_BODIES][3], cap_p1[PARC_SIM_MAX_BODIES][3], cap_radius[PARC_SIM_MAX_BODIES];
    uint32_t self_mask[PARC_SIM_MAX_BODIES];
} parc_sim_model_t;

/* One control step for n_envs environments: n_substeps semi-implicit Euler substeps of length h with PD
 * targets = clamp(action, action_low, action_high).  model is a DEVICE pointer to a parc_sim_model_t.
 * root_state [N,13], dof_state [N,D,2], rigid_body_state [N,B,13], contact_forces [N,B,3] (mean over the
 * substeps, env frame), env_offsets [N,3], action [N,D], action_low/high [D]. */
int parc_sim_step(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                  float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets,
                  const float *action, const float *action_low, const float *action_high, int n_substeps, float h);

/* The same step, followed by IGEnv._update_time (envs/ig_env.py:862-865) inside the launch: timestep_buf[e] += 1 (int32),
 * time_buf[e] = timestep_buf[e] * step_dt.  */
int parc_sim_step_tick(void *stream, const parc_sim_model_t *model, parc_terrain_t terrain, int n_envs, float *root_state,
                       float *dof_state, float *rigid_body_state, float *contact_forces, const float *env_offsets,
                       const float *action, const float *action_low, const float *action_high, int n_substeps, float h,
                       int32_t *timestep_buf, float *time_buf, float step_dt);

/* Recompute rigid_body_state (poses, velocities) from root_state / dof_state for the listed envs and zero
 * their contact forces: what the reference gets from refresh_rigid_body_state_tensor after a reset
 * (envs/ig_env.py:850-860).  env_ids int64 device pointer, NULL = all. */
int parc_sim_refresh_bodies(void *stream, const parc_sim_model_t *model, int n_envs, const int64_t *env_ids, int n_sel,
                            const float *root_state, const float *dof_state, float *rigid_body_state, float *contact_forces);

/* Same for every env whose mask[e] != 0 (device-side reset, no index list). */
int parc_sim_refresh_bodies_masked(void *stream, const parc_sim_model_t *model, int n_envs, const int32_t *mask,
                                   const float *root_state, const float *dof_state, float *rigid_body_state, float *contact_forces);

int parc_sim_abi(void);

#ifdef __cplusplus
}
#endif
#endif
