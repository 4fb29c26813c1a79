/*
 * parc_hip.h -- C-ABI of libparc_hip.so: the MI355X (gfx950) hot path of the PARC motion tracker.
 *
 * The reference (ZhengmaoHe/PARC) has no FFI of its own: its hot path sits behind Python classes
 * (envs/ig_parkour/ig_parkour_env.py IGParkourEnv, learning/dm_ppo_agent.py DMPPOAgent) that call
 * Isaac Gym binaries and chains of torch ops.  This library replaces exactly those call sites; each
 * entry point names the reference function(s) it stands in for (paths relative to the reference
 * root).  parc_amd/ binds it with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions:
 *   - plain pointers and sizes only; every data pointer is a DEVICE pointer unless the name says host;
 *   - all arithmetic fp32, indices int64 where the reference uses LongTensor, flags int32;
 *   - `stream` is a hipStream_t passed as void*; functions enqueue work and return, they never
 *     synchronise, allocate or free;  return 0 on success, a negative PARC_E* code on bad arguments,
 *     a positive hipError_t if a launch failed;
 *   - quaternions are xyzw.
 */
#ifndef PARC_HIP_H
#define PARC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PARC_MAX_BODIES 16
#define PARC_MAX_DOFS 64
#define PARC_MAX_TAR_STEPS 6
#define PARC_MAX_KEY_BODIES 8

#define PARC_OK 0
#define PARC_EINVAL (-1)
#define PARC_EUNSUPPORTED (-2)

/* anim/kin_char_model.py:11-15 JointType */
#define PARC_JOINT_ROOT 0
#define PARC_JOINT_HINGE 1
#define PARC_JOINT_SPHERICAL 2
#define PARC_JOINT_FIXED 3

/* envs/base_env.py:12-16 DoneFlags */
#define PARC_DONE_NULL 0
#define PARC_DONE_FAIL 1
#define PARC_DONE_SUCC 2
#define PARC_DONE_TIME 3

/* Kinematic tree, host struct passed by value.  anim/kin_char_model.py:142-178 (KinCharModel.init). */
typedef struct {
    int32_t num_bodies;                         /* 15 for data/assets/humanoid.xml */
    int32_t dof_size;                           /* 28 */
    int32_t max_depth;
    int32_t parent[PARC_MAX_BODIES];            /* -1 for the root */
    int32_t joint_type[PARC_MAX_BODIES];
    int32_t dof_idx[PARC_MAX_BODIES];
    int32_t depth[PARC_MAX_BODIES];
    float local_translation[PARC_MAX_BODIES][3];
    float local_rotation[PARC_MAX_BODIES][4];
    float joint_axis[PARC_MAX_BODIES][3];
} parc_char_model_t;

/*
 * Clip database in HBM.  anim/motion_lib.py:204-380 (MotionLib._load_motions) keeps seven flat frame
 * arrays; here one frame is ONE 16-byte-aligned row so a query gathers two contiguous rows:
 *   [0, 4B)               quaternions: root_rot, joint_rot[0..J)      (B = num_bodies quats)
 *   [off_pos, +3)         root_pos
 *   [off_contacts, +B)    contacts
 *   [off_root_vel, +3) [off_root_ang_vel, +3) [off_dof_vel, +D)   (taken from frame idx0 only)
 */
typedef struct {
    int32_t num_motions, num_bodies, dof_size, row_stride; /* row_stride in floats, multiple of 4 */
    int32_t off_pos, off_contacts, off_root_vel, off_root_ang_vel, off_dof_vel;
    const int32_t *num_frames;   /* [M] */
    const int32_t *start_idx;    /* [M] */
    const float *length;         /* [M] seconds */
    const int32_t *loop_mode;    /* [M] 0 CLAMP, 1 WRAP  (anim/motion_lib.py:11-13) */
    const float *pos_delta;      /* [M,3] */
    const float *frames;         /* [total_frames, row_stride] */
} parc_motion_lib_t;

/* util/terrain_util.py:21-40 SubTerrain: hf[X,Y] row-major, cell centre = min_point + ij*dxdy */
typedef struct {
    const float *hf;
    int32_t dim_x, dim_y;
    float min_x, min_y, dx, dy;
} parc_terrain_t;

/* Scalars of the tracker env config (PARC/tracker_config/dm_env_default.yaml), host struct by value. */
typedef struct {
    int32_t num_tar_steps;                       /* tar_obs_steps: 6 */
    float tar_dt[PARC_MAX_TAR_STEPS];            /* timestep * tar_obs_steps, seconds */
    int32_t num_key_bodies;
    int32_t key_body_ids[PARC_MAX_KEY_BODIES];
    float joint_err_w[PARC_MAX_BODIES];          /* [J] */
    float dof_err_w[PARC_MAX_DOFS];              /* [D] */
    float contact_w[PARC_MAX_BODIES];            /* [B] */
    float reward_w[5];                           /* pose, vel, root_pos, root_vel, key_pos (normalised) */
    float rel_deepmimic_w;
    float pose_termination_dist[PARC_MAX_BODIES];/* [J] */
    int32_t pose_termination, enable_early_termination, track_root;
    float root_pos_termination_dist, root_rot_termination_angle, termination_height;
    int32_t num_contact_bodies;
    int32_t contact_body_mask[PARC_MAX_BODIES];  /* 1 = body may touch the ground */
    float episode_length, contact_eps, min_obs_h, max_obs_h;
    int32_t num_ray_points;                      /* 441 */
    int32_t obs_dim;                             /* 1312 */
    float task1_w, task2_w, target_radius;       /* task reward terms, logged only while rel_task_w == 0 */
    float target_future_min, target_future_max;  /* dm.target_xy_future_time_min/max (PARC_POST_TARGETS) */
    /* reward variants (ig_parkour_env.py:1284-1285,1323-1339; mgdm_dm_util.py:343-350): track_root_h = 0 drops the height from the
     * root position error, use_contact_info = 0 drops the contact penalty from the reward. */
    int32_t track_root_h, use_contact_info;
    /* global_obs (ig_char_env.py:585-590, mgdm_dm_util.py:476-500): observation in world axes instead of the heading frame.
     * track_root = 0 (the field above) also changes the reward: xy of the root error dropped, root rotation / velocities / key bodies
     * compared in each character's own heading frame (mgdm_dm_util.py:343-360). */
    int32_t global_obs;
} parc_track_cfg_t;

/*
 * Per-env device buffers.  The first four are the Isaac Gym state tensors the reference wraps at
 * envs/ig_env.py:764-780 (same row layout: pos3 quat4 linvel3 angvel3).
 */
typedef struct {
    int32_t num_envs;
    const float *root_state;        /* [N,13] */
    const float *dof_state;         /* [N,D,2] pos,vel interleaved */
    const float *rigid_body_state;  /* [N,B,13] */
    const float *contact_forces;    /* [N,B,3] */
    const float *env_offsets;       /* [N,3]   ig_parkour_env.py:496-501 */
    const int64_t *motion_ids;      /* [N]     dm_env.py:98 */
    const float *motion_time_offsets; /* [N]   dm_env.py:100 */
    const float *motion_xy_offset;  /* [N,2] = motion_offsets[motion_id, terrain_id]  (dm_env.py:604-615) */
    const float *time_buf;          /* [N] */
    const float *target_xy;         /* [N,2]   ig_parkour_env.py:791 (task reward only) */
    /* outputs */
    float *ref_root_pos, *ref_root_rot, *ref_root_vel, *ref_root_ang_vel; /* [N,3] [N,4] [N,3] [N,3] */
    float *ref_joint_rot, *ref_dof_vel, *ref_dof_pos;                     /* [N,J,4] [N,D] [N,D] */
    float *ref_contacts, *ref_body_pos;                                   /* [N,B] [N,B,3] */
    float *obs;                     /* [N,obs_dim]; columns [obs_dim-P, obs_dim) belong to parc_refresh_obs_hfs */
    float *reward;                  /* [N] */
    float *reward_terms;            /* [9,N]: pose_r vel_r root_pos_r root_vel_r key_pos_r contact_penalty task_r1 task_r2 total_task_r */
    int32_t *done;                  /* [N]  DoneFlags after the motion-end override (dm_env.py:782) */
    int32_t *done_kind;             /* [N]  0 none, 1 failed, 2 ended without failing (feeds parc_update_fail_rates) */
    /* device-side reset (no host round trip): see PARC_POST_MASKED / PARC_POST_INIT_CHAR; both may be NULL otherwise */
    const int32_t *env_mask;        /* [N]  nonzero = env takes part in a PARC_POST_MASKED launch */
    const float *init_noise_xy;     /* [N,2] or NULL: added to the root xy written by PARC_POST_INIT_CHAR (already scaled) */
    /* PARC_POST_TARGETS: DeepMimicEnv._update_motion_targets (dm_env.py:617-654) inside the launch */
    float *next_target_time;        /* [N]   time at which an env draws its next xy target */
    const float *target_rand;       /* [N,3] uniforms in [0,1): look-ahead time, and a Box-Muller pair for the 0.05 m target noise */
    /* Per-env values that belong to observation columns OUTSIDE the fused row layout (optional variants of _compute_obs,
     * ig_parkour_env.py:1212-1224 and ig_char_env.py:618-620; gathered into the handed-out row by parc_assemble_obs):
     * [N,4] = root height (root_pos z), the xy target localised to the root (rotate_2d_vec(target_xy - root_xy, -heading)), 0.
     * Written by every PARC_POST_OBS launch when not NULL. */
    float *obs_aux;
    /* A launch on a ROW RANGE [e0, e0 + n) of larger allocations (a sub-env: every pointer above advanced by e0 rows, num_envs = n)
     * keeps the term-major layout of reward_terms by naming the allocation's row length here; 0 = num_envs. */
    int32_t reward_terms_stride;
} parc_env_buffers_t;

/* ---- K5: local heightmap ---------------------------------------------------------------------
 * RefCharEnv._refresh_ray_obs_hfs  envs/ig_parkour/mgdm_dm_util.py:158-179
 * + terrain_util.get_local_hf_from_terrain util/terrain_util.py:1329-1346, SubTerrain.get_grid_index :113-126.
 * root_pos_xyz [N,3] is the GLOBAL position (env offset added), heading [N] = calc_heading(root_rot).
 * out row e starts at out + e*out_stride (floats); P values per row: clamp(hf - z, min_h, max_h). */
int parc_refresh_ray_obs_hfs(void *stream, int n_envs, const float *ray_xy, int n_points,
                             const float *root_pos_xyz, const float *heading, parc_terrain_t terrain,
                             float min_h, float max_h, float *out, int64_t out_stride);

/* IGParkourEnv._refresh_obs_hfs  envs/ig_parkour/ig_parkour_env.py:636-656: same, but straight from the
 * simulator's root_state [N,13] and env_offsets [N,3] (heading computed in-kernel, no temporaries). */
int parc_refresh_obs_hfs(void *stream, int n_envs, const float *ray_xy, int n_points, const float *root_state,
                         const float *env_offsets, parc_terrain_t terrain, float min_h, float max_h, float *out,
                         int64_t out_stride);

/* ---- K1/K4/K2: KinCharModel.dof_to_rot / rot_to_dof / forward_kinematics  anim/kin_char_model.py:478-541 */
int parc_dof_to_rot(void *stream, parc_char_model_t model, int n, const float *dof, float *joint_rot);
int parc_rot_to_dof(void *stream, parc_char_model_t model, int n, const float *joint_rot, float *dof);
int parc_forward_kinematics(void *stream, parc_char_model_t model, int n, const float *root_pos, const float *root_rot,
                            const float *joint_rot, float *body_pos, float *body_rot);

/* The pose chain of a batch of frames and its adjoint (what stage 2's motion optimiser differentiates: util/torch_util.py:414-419
 * exp_map_to_quat, anim/kin_char_model.py:478-541 dof_to_rot + forward_kinematics, tools/motion_opt/motion_optimization.py:203-213).
 * forward: root_pos [n,3], root_exp [n,3] (exponential map), dof [n,D] -> root_quat [n,4], joint_rot [n,J,4], body_pos [n,B,3],
 * body_rot [n,B,4].  backward: the cotangents of those four outputs -> g_root_pos [n,3], g_root_exp [n,3], g_dof [n,D] (overwritten).
 * Bodies must be stored parents-first (MJCF depth-first order). */
int parc_pose_chain_forward(void *stream, parc_char_model_t model, int n, const float *root_pos, const float *root_exp, const float *dof,
                            float *root_quat, float *joint_rot, float *body_pos, float *body_rot);
int parc_pose_chain_backward(void *stream, parc_char_model_t model, int n, const float *root_exp, const float *dof,
                             const float *g_root_quat, const float *g_joint_rot, const float *g_body_pos, const float *g_body_rot,
                             float *g_root_pos, float *g_root_exp, float *g_dof);

/* Sample points of the bodies in the world frame, world [T,P,3] = body_pos[t, owner[p]] + rotate(body_rot[t, owner[p]], local[p])
 * (what the terrain losses feed to parc_points_hf_sdf: util/terrain_util.py:1895-1951, tools/motion_opt/motion_optimization.py:241-247),
 * and its adjoint g_world -> g_body_pos [T,B,3], g_body_rot [T,B,4]; a body's points are contiguous, start [B+1] are their offsets. */
int parc_body_points_world(void *stream, int n_frames, int num_bodies, int num_points, const float *body_pos, const float *body_rot,
                           const float *local, const int32_t *owner, float *world);
int parc_body_points_world_grad(void *stream, int n_frames, int num_bodies, int num_points, const float *body_rot, const float *local,
                                const int32_t *start, const float *g_world, float *g_body_pos, float *g_body_rot);

/* torch_util.quat_diff_angle (util/torch_util.py:427-431) for n quaternion pairs and its adjoint (g_q0, g_q1 [n,4] overwritten):
 * the rotation-error terms of stage 2's optimiser (motion_optimization.py:203-221). */
int parc_quat_diff_angle(void *stream, int64_t n, const float *q0, const float *q1, float *angle);
int parc_quat_diff_angle_grad(void *stream, int64_t n, const float *q0, const float *q1, const float *g_angle, float *g_q0, float *g_q1);

/* Frame-to-frame terms of stage 2's motion loss per (frame, body) and their adjoint (motion_optimization.py:215-224,346-362):
 * partial [3, T, B] = smoothness |v - v_src|^2 + r, sliding pseudo-Huber terms (masked by `keep`, weighted by `pair_contact`), jerk
 * max(|third difference of body_pos| - jerk_limit, 0); v = body_pos[t+1] - body_pos[t]; rot_err_sq / src_vel / keep / pair_contact are
 * [T-1, B(,3)] (row T-1 of the [T, B] index space is not read).  The caller sums the partials; the adjoint takes the cotangents of the
 * three sums (device, 3 floats) and overwrites g_body_pos [T,B,3] and g_rot_err_sq [T-1,B]. */
int parc_temporal_terms(void *stream, int n_frames, int num_bodies, const float *body_pos, const float *rot_err_sq, const float *src_vel,
                        const float *keep, const float *pair_contact, float c, float c2, float jerk_limit, float *partial);
int parc_temporal_terms_grad(void *stream, int n_frames, int num_bodies, const float *body_pos, const float *rot_err_sq, const float *src_vel,
                             const float *keep, const float *pair_contact, float c, float c2, float jerk_limit, const float *cotangents,
                             float *g_body_pos, float *g_rot_err_sq);

/* ---- K3: MotionLib.calc_motion_frame  anim/motion_lib.py:80-112 (contact_info=True)
 * outputs AoS: root_pos[Q,3] root_rot[Q,4] root_vel[Q,3] root_ang_vel[Q,3] joint_rot[Q,J,4] dof_vel[Q,D] contacts[Q,B] */
int parc_calc_motion_frame(void *stream, parc_motion_lib_t mlib, int n_queries, const int64_t *motion_ids,
                           const float *motion_times, float *root_pos, float *root_rot, float *root_vel,
                           float *root_ang_vel, float *joint_rot, float *dof_vel, float *contacts);

/* ---- clip database build: MotionLib._load_motions  anim/motion_lib.py:264-290,405-423 and
 * KinCharModel.compute_frame_dof_vel  anim/kin_char_model.py:543-581 for all clips at once.
 * frames [F,6+D] (root pos, root exp map, dofs), contacts [F,B] or NULL, frame_clip [F] clip index of
 * each frame, clip_fps [M]; mlib carries num_frames/start_idx and the row layout; rows [F,row_stride] out. */
int parc_motion_lib_build(void *stream, parc_char_model_t model, parc_motion_lib_t mlib, int total_frames,
                          const float *frames, const float *contacts, const int32_t *frame_clip, const float *clip_fps,
                          float *rows);

/* ---- fused post-physics pass: one launch per env step ------------------------------------------
 * IGEnv._post_physics_step  envs/ig_env.py:839-848 for the tracker:
 *   DeepMimicEnv._update_ref_motion        dm_env.py:570-595          (K3 + tile shift + K2 + K4)
 *   IGParkourEnv._compute_obs              ig_parkour_env.py:1054-1244 (K1 K2 K6 K7 K8, columns [0, obs_dim-P))
 *   IGParkourEnv._update_reward            ig_parkour_env.py:1275-1339,1399-1404 (K9)
 *   RefCharEnv.update_done / DeepMimicEnv.update_done  mgdm_dm_util.py:205-230,392-460, dm_env.py:746-783 (K10)
 * env_ids (int64, device) selects a subset (reset path: ig_parkour_env.py:1033-1036); NULL = all envs.
 * what: bit0 ref-state update, bit1 observations, bit2 reward+done, bit3 fuse the local heightmap (K5,
 * ig_parkour_env.py:636-656) into the same launch: the whole obs row is then written at once and
 * parc_refresh_obs_hfs is not needed (requires bit1; ray_xy [P,2] device pointer, else may be NULL). */
#define PARC_POST_REF 1
#define PARC_POST_OBS 2
#define PARC_POST_REWARD_DONE 4
#define PARC_POST_HF 8
/* bit4: only envs with buf.env_mask[e] != 0 are processed (all N are scanned; env_ids must be NULL): the reset of
 * finished envs without the nonzero()/index round trip of ig_env.py:100-121.
 * bit5 (with bit0): RefCharEnv._char_state_init_from_ref + add_noise_to_char_state (mgdm_dm_util.py:119-136): the
 * reference state at the env's clip time is ALSO written into root_state / dof_state rows of the env (the buffers are
 * written through their const pointers in this mode only). */
#define PARC_POST_MASKED 16
#define PARC_POST_INIT_CHAR 32
/* bit6: envs whose time reached next_target_time draw a new xy target = clip root position at (clip time + U[future_min,
 * future_max]) + N(0, 0.05 m), written to target_xy / next_target_time (target_xy is written through its const pointer in
 * this mode only) before the task reward terms read it. */
#define PARC_POST_TARGETS 64
/* bit7: the rows follow generated plans (the motion-generator sub-env, mgdm_env.py): clip time = motion_time_offsets[e] alone (the
 * global plan clock, mgdm_env.py:476-480 - not env time + offset), and no end-of-clip termination (DeepMimicEnv.update_done's rule
 * dm_env.py:746-783 does not apply; the sub-env's termination is RefCharEnv.update_done, mgdm_dm_util.py:205-230). */
#define PARC_POST_PLAN_CLOCK 128
/* every documented bit; parc_track_post_step returns PARC_EINVAL for any other bit of `what` */
#define PARC_POST_ALL 255
/* Two kernels since round 3: bit0 (the reference STATE: ref_* buffers, bit5's character state) is `ref_state_kernel`, enqueued first;
 * bits 1-3 / 6 the fused `track_post_kernel`, whose reward wave samples the reference pose itself.  They write disjoint outputs and
 * read nothing of each other.  A call with bit0 only (a restart's first launch) enqueues the small kernel alone. */
int parc_track_post_step(void *stream, parc_char_model_t model, parc_motion_lib_t mlib, parc_terrain_t terrain,
                         parc_track_cfg_t cfg, parc_env_buffers_t buf, const int64_t *env_ids, int n_sel, int what,
                         const float *ray_xy);

/* The same call with two hipEvent_t (created by the caller with timing enabled) bound to the fused kernel's dispatch
 * (hipExtLaunchKernel): after the stream has passed it, hipEventElapsedTime(start, stop) is the duration of that one launch by the
 * dispatch's own time stamps.  Results are identical to parc_track_post_step; bench.py uses it to time the launch where the rollout
 * step issues it (an event pair recorded AROUND a launch adds ~5 us of event handling on this runtime).  Not capturable in a graph. */
int parc_track_post_step_timed(void *stream, parc_char_model_t model, parc_motion_lib_t mlib, parc_terrain_t terrain,
                               parc_track_cfg_t cfg, parc_env_buffers_t buf, const int64_t *env_ids, int n_sel, int what,
                               const float *ray_xy, void *start_event, void *stop_event);

/* Observation rows of the non-default layouts of IGParkourEnv._compute_obs (ig_parkour_env.py:1054-1244): global_root_height_obs
 * prepends the root height, enable_tar_obs = False / use_contact_info = False leave column blocks out, has_target_xy_obs and the
 * replan timer append columns behind the heightmap.  out[e, c] = V[e, col_map[c]] where the virtual row V[e] is
 * [ obs[e, 0..obs_dim) | aux[e, 0..4) | *scalar ] (col_map[c] in [0, obs_dim + 5); aux / scalar may be NULL if unused).
 * col_map: int32 [out_dim] on the device.  env_ids (int64 device pointer, n_sel entries) restricts the pass to those rows - the rows
 * a reset touches (ig_env.py:100-121: reset(env_ids) rewrites obs_buf[env_ids] only) -, NULL = all n_envs rows.  One coalesced pass,
 * no temporaries. */
int parc_assemble_obs(void *stream, int n_envs, const float *obs, int obs_dim, const float *aux, const float *scalar,
                      const int32_t *col_map, float *out, int out_dim, const int64_t *env_ids, int n_sel);

/* Reset bookkeeping of DeepMimicEnv._reset_envs / IGEnv._reset_envs (dm_env.py:517-568, ig_env.py:100-121,693-721) for the
 * envs whose mask is set, from per-env candidate samples new_*[N] (drawn for every env, used where masked):
 * motion_ids/terrain ids/time offsets/tile offset are replaced, timestep/time/done/next-target-time cleared, ep_num += 1.
 * motion_offsets is the [M, terrains_per_motion, 2] tile table. */
int parc_reset_apply(void *stream, int n_envs, const int32_t *mask, const int64_t *new_motion_ids, const int64_t *new_terrain_ids,
                     const float *new_time_offsets, const float *motion_offsets, int terrains_per_motion, int64_t *motion_ids,
                     int64_t *motion_terrain_ids, float *motion_time_offsets, float *motion_xy_offset, int32_t *timestep_buf,
                     float *time_buf, int32_t *done, float *next_target_time, int64_t *ep_num);

/* The same reset with the sampling inside (DeepMimicEnv._reset_ref_motion dm_env.py:517-568, MotionLib.sample_motions /
 * sample_time anim/motion_lib.py:48-63, add_noise_to_char_state mgdm_dm_util.py:128-136): envs with done_flags[e] != 0 restart;
 * mask[e] (out) = that condition.  uniforms [5, n_envs] in [0,1): row 0 picks the clip (probability ~ motion_weights[m] *
 * max(fail_rates[m], min_weight); fail_rates NULL = weights alone), row 1 the tile copy, row 2 the start phase (time = u * clip
 * length), rows 3-4 the xy start noise (2u-1) * noise_scale written to init_noise_xy.  cdf_workspace: n_motions floats. */
int parc_reset_sample_apply(void *stream, int n_envs, const int32_t *done_flags, int32_t *mask, const float *uniforms, int n_motions,
                            const float *motion_weights, const float *fail_rates, float min_weight, const float *motion_lengths,
                            const float *motion_offsets, int terrains_per_motion, float noise_scale, float *cdf_workspace,
                            int64_t *motion_ids, int64_t *motion_terrain_ids, float *motion_time_offsets, float *motion_xy_offset,
                            int32_t *timestep_buf, float *time_buf, int32_t *done, float *next_target_time, int64_t *ep_num,
                            float *init_noise_xy);

/* DeepMimicEnv.update_done's per-done-env Python loop (dm_env.py:758-772): EMA of per-clip failure rates,
 * applied in increasing env order exactly as the reference's loop does. */
/* The tail of an env step in one launch of heterogeneous workgroups: the fail-rate update above (one workgroup per clip) and, with
 * what = PARC_POST_REF (| PARC_POST_PLAN_CLOCK), the per-step publication of the reference state - ref_* buffers of `buf`,
 * DeepMimicEnv._update_ref_motion dm_env.py:570-595 - by further workgroups that run on the CUs the fail-rate walk leaves idle.  A
 * caller that uses it leaves PARC_POST_REF out of the parc_track_post_step of that step. */
int parc_step_tail(void *stream, parc_char_model_t model, parc_motion_lib_t mlib, parc_env_buffers_t buf, int what, int n_motions,
                   const int32_t *done_kind, float ema_w, float *fail_rates);
int parc_update_fail_rates(void *stream, int n_envs, int n_motions, const int64_t *motion_ids, const int32_t *done_kind,
                           float ema_w, float *fail_rates);

/* ---- K16/K17: rl_util.compute_td_lambda_return  learning/rl_util.py:6-29  (time-major [T,N]) and the
 * advantage normalisation of DMPPOAgent._build_train_data  learning/dm_ppo_agent.py:393-403.
 * workspace: >= 3*1024 doubles (device), zeroing not required. */
int parc_td_lambda_return(void *stream, int T, int N, const float *reward, const float *next_vals, const int32_t *done,
                          float discount, float td_lambda, float *ret);
int parc_adv_normalize(void *stream, int n, const float *ret, const float *vals, const float *rand_action_mask,
                       float clip, float *norm_adv, float *mean_std_out /* [2] */, double *workspace);

/* ---- PPO loss + gradient in one pass: PPOAgent._compute_loss  learning/ppo_agent.py:186-330 (diagonal Gaussian policy with a
 * state-independent log-std; masked means over rand_action_mask == 1; clip / action-bound / entropy / mean-regulariser terms;
 * L2 or L1 critic loss; the "large critic loss" guard that stops the actor gradient).
 * mean, norm_a [B,A]; logstd [A]; old_logp, adv, mask, pred, tar_val [B].
 * Outputs: g_mean [B,A] = dloss/dmean, g_logstd [A], g_pred [B];
 * out[11] = loss, critic_loss, actor_loss, clip_frac, imp_ratio, action_bound_loss, entropy, reg_loss, cnt, (2 scale factors).
 * workspace: parc_ppo_workspace_floats(B) floats. */
typedef struct {
    float clip_ratio, bound_w, entropy_w, reg_w, critic_w, large_critic_loss;
    int32_t critic_l1;
} parc_ppo_cfg_t;
int parc_ppo_loss(void *stream, int B, int A, const float *mean, const float *logstd, const float *norm_a, const float *old_logp,
                  const float *adv, const float *mask, const float *pred, const float *tar_val, parc_ppo_cfg_t cfg, float *g_mean,
                  float *g_logstd, float *g_pred, float *out, float *workspace);
int parc_ppo_workspace_floats(int B);
/* The same loss on packed per-sample records rec[B, rec_stride] = [norm_action (A) | a_logp | adv | rand_action_mask | tar_val | pad]
 * (rec_stride >= A + 4): one row gather per minibatch delivers all per-sample inputs of PPOAgent._compute_loss. */
int parc_ppo_loss_packed(void *stream, int B, int A, const float *mean, const float *logstd, const float *rec, int rec_stride,
                         const float *pred, parc_ppo_cfg_t cfg, float *g_mean, float *g_logstd, float *g_pred, float *out,
                         float *workspace);

/* ---- K15 experience record: ExperienceBuffer.record  learning/experience_buffer.py:55-59 for a group of buffers at once.
 * Field f copies src [N, row_bytes] into row *head of dst [T, N, row_bytes] (row_bytes a multiple of 4; 16-byte vector copies
 * when sizes and pointers allow).  convert = 1: src is int64 [N], dst int32 [T, N] (the reference's ep_num buffer), row_bytes = 8;
 * convert = 2: src is ONE 4-byte value written to every env of the row (row_bytes = 4; base_agent records the plan clock this way).
 * head: device int64 scalar (so the launch can sit inside a captured graph); fields: HOST array of n_fields <= 12 descriptors,
 * read during the call (they travel as kernel arguments). */
typedef struct {
    const void *src;
    void *dst;
    int32_t row_bytes;
    int32_t convert;
} parc_record_field_t;
int parc_record_step(void *stream, int n_envs, const int64_t *head, int n_fields, const parc_record_field_t *fields);

/* The random numbers of one rollout step in one launch (counter-based Philox4x32-10 keyed by `seed`): n_uniform floats in [0, 1) and
 * n_normal floats ~ N(0, 1).  state: two uint64 on the device - [0] the step counter, advanced by the launch; [1] a ticket, zero
 * between launches.  Same (seed, counter) -> same numbers on every run.  tick_cell (may be NULL): a device int64 that the launch - the
 * first of a rollout step - moves on by one modulo tick_mod: ExperienceBuffer.inc (experience_buffer.py:41-44) for a write row kept on
 * the device (the `head` of parc_record_step). */
int parc_rng_step(void *stream, uint64_t seed, uint64_t *state, float *uniform_out, int64_t n_uniform, float *normal_out, int64_t n_normal,
                  int64_t *tick_cell, int tick_mod);

/* K20 in two passes over flat buffers (MPOptimizer.step, learning/mp_optimizer.py:20-40: clip_grad_norm_ then SGD with momentum):
 * norm = |grad|_2 (fixed summation order), coef = min(max_norm / (norm + 1e-6), 1) (max_norm <= 0: no clipping), g' = coef grad
 * (+ weight_decay * p), momentum_buf = momentum * momentum_buf + g', params -= lr * momentum_buf.  grad is left as it was.
 * grad 16-byte aligned; workspace: parc_sgd_workspace_floats() floats; norm_out (may be NULL): the gradient norm. */
int64_t parc_sgd_workspace_floats(void);
int parc_sgd_momentum_step(void *stream, int64_t n, float *params, const float *grad, float *momentum_buf, float max_norm, float lr,
                           float momentum, float weight_decay, float *workspace, float *norm_out);

/* ---- K12: Normalizer.normalize  learning/normalizer.py:60-63 in one pass: out = clamp((x - mean) / std, -clip, clip).
 * x, out [rows, dim] row-major, mean / std [dim]; dim a multiple of 4, 16-byte aligned pointers; out may alias x. */
int parc_normalize_clamp(void *stream, int64_t rows, int dim, const float *x, const float *mean, const float *stdv, float clip, float *out);

/* ---- K22: Normalizer.record  learning/normalizer.py:28-34: acc[0,:] += sum over rows of x, acc[1,:] += sum over rows of x*x
 * in one pass with a fixed summation order.  x [rows, dim] row-major, acc [2, dim], dim a multiple of 4, 16-byte aligned;
 * workspace: parc_moments_workspace_floats(rows, dim) floats of scratch (caller-owned). */
int64_t parc_moments_workspace_floats(int64_t rows, int dim);
int parc_moments_accumulate(void *stream, int64_t rows, int dim, const float *x, float *acc, float *workspace);
/* The three passes a rollout step makes over its observation rows in one: norm_out = parc_normalize_clamp(x) (bit-identical);
 * copy_dst (may be NULL): x is also written into time row *copy_row (device int64) of a [T, rows, dim] buffer - ExperienceBuffer.record
 * experience_buffer.py:55-59; acc (may be NULL): parc_moments_accumulate(x) with the same partial rows and summation order
 * (workspace as there).  All pointers 16-byte aligned, dim a multiple of 4. */
int parc_obs_ingest(void *stream, int64_t rows, int dim, const float *x, const float *mean, const float *stdv, float clip, float *norm_out,
                    float *copy_dst, const int64_t *copy_row, float *acc, float *workspace);

/* ---- K14: the part of PPOAgent._decide_action (learning/ppo_agent.py:87-119) after the actor MLP: sample / mode by the
 * exploration mask, log-probability, un-normalised action.  mean, noise, action [n, A]; logstd, a_mean, a_std [A]; explore,
 * logp [n]. */
int parc_action_head(void *stream, int n, int A, const float *mean, const float *logstd, const float *noise, const float *explore,
                     const float *a_mean, const float *a_std, float *action, float *logp);
/* The same, and in the same launch what ExperienceBuffer.record stores of this moment (base_agent's _record_data_pre_step,
 * dm_ppo_agent.py:289-299): action / a_logp / rand_action_mask (= explore) and the env's contact forces forces_src [n, n_forces] into
 * time row *head (device int64) of rec_action [T, n, A], rec_logp [T, n], rec_mask [T, n], rec_forces [T, n, n_forces].  A <= 32. */
int parc_action_head_record(void *stream, int n, int A, const float *mean, const float *logstd, const float *noise, const float *explore,
                            const float *a_mean, const float *a_std, float *action, float *logp, float *rec_action, float *rec_logp,
                            float *rec_mask, const float *forces_src, float *rec_forces, int n_forces, const int64_t *head);

/* ---- next rows: terrain geometry either side of the tracker -----------------------------------------------
 * terrain_util.points_hf_sdf  util/terrain_util.py:1835-1893 (+ points_boxes_sdf :1774-1804, geom_util.sdBox/sdRoundBox
 * util/geom_util.py:113-143): signed distance of points [batch, n_points, 3] to the columns of hf [batch, dim_x, dim_y].
 * Cell (i, j) of batch b is the box centred at (x_points[i] + min_box_center[b,0], y_points[j] + min_box_center[b,1]) with
 * half extents (half_x, half_y); vertically [base_z, hf] or, inverted != 0, [hf, -base_z] with the result negated.
 * radius > 0 = rounded boxes (sd - radius), <= 0 = plain boxes.  out [batch, n_points]; out_cell (may be NULL) [batch, n_points]:
 * flat index i * dim_y + j of the first column attaining the minimum (what a caller needs to differentiate the distance). */
int parc_points_hf_sdf(void *stream, int batch, int n_points, int dim_x, int dim_y, const float *points, const float *hf,
                       const float *min_box_center, const float *x_points, const float *y_points, float half_x, float half_y,
                       float base_z, int inverted, float radius, float *out, int32_t *out_cell);

/* The adjoint of parc_points_hf_sdf with respect to the points: g_points [B,N,3] = g_out [B,N] * d(out)/d(point) for the column the
 * forward call reported in out_cell (what the reference obtains from autograd through util/terrain_util.py:1835-1893). */
int parc_points_hf_sdf_grad(void *stream, int batch, int n_points, int dim_x, int dim_y, const float *points, const float *hf,
                            const float *min_box_center, const float *x_points, const float *y_points, float half_x, float half_y,
                            float base_z, int inverted, const int32_t *cell, const float *g_out, float *g_points);

/* Backward of a Linear + ReLU layer between its two GEMMs (the derivative of learning/nets/fc_3layers_2048units.py:4-22): in one pass
 * gy[r, c] <- gy[r, c] * (y[r, c] > 0) (in place) and db[c] <- sum_r of the result (overwritten, fixed summation order).
 * gy, y: [rows, dim] row-major, dim % 4 == 0, 16-byte aligned; workspace: parc_relu_bwd_workspace_floats(rows, dim) floats. */
int64_t parc_relu_bwd_workspace_floats(int64_t rows, int dim);
int parc_relu_bwd_bias_grad(void *stream, int64_t rows, int dim, float *gy, const float *y, float *db, float *workspace);
/* out[c] <- sum_r w[r] * x[r, c] (x [rows, dim] row-major, dim % 4 == 0, 16-byte aligned; workspace as above): the weight gradient of
 * a Linear layer with one output (the value head `_critic_out`, learning/ppo_model.py:14-22), fixed summation order. */
int parc_weighted_colsum(void *stream, int64_t rows, int dim, const float *x, const float *w, float *out, float *workspace);

int parc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PARC_HIP_H */
