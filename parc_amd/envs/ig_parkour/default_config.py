"""Default tracker configuration as Python data (what PARC/tracker_config/dm_env_default.yaml and
dm_agent_default.yaml configure in the reference).  Used by tests, smoke and bench; real runs pass YAML files
through envs.env_builder / learning.agent_builder exactly like the reference."""
import copy

_ENV = {
    "env_name": "ig_parkour",
    "env": {
        "camera_mode": "track",
        "char_file": None,                  # filled by default_env_config()
        "contact_bodies": [],
        "contact_detection_eps": 1.0e-05,
        "contact_weights": [5.0] * 15,
        "control_freq": 30,
        "control_mode": "pd",
        "debug_visuals": False,
        "demo_mode": False,
        "dm": {
            "fail_rate_quantiles": [0.25, 0.5, 0.75, 0.9, 0.95, 0.99],
            "heightmap": {"horizontal_scale": 0.4, "padding": 0.4},
            "min_motion_weight": 0.01,
            "has_motion_classes": False,
            "motion_classes": [],
            "motion_file": None,
            "target_xy_future_time_max": 5.0,
            "target_xy_future_time_min": 1.0,
            "terrain_build_mode": "square",
            "terrain_save_path": None,
            "terrains_per_motion": 1,
        },
        "enable_early_termination": True,
        "enable_replan_timer_obs": True,
        "env_spacing": 2.0,
        "env_style": "square",
        "episode_length": 10.0,
        "fraction_dm_envs": 1.0,
        "global_obs": False,
        "global_root_height_obs": False,
        "has_target_xy_obs": False,
        "init_pose": [0, 0, 0.882416] + [0] * 31,
        "joint_err_w": [1.0, 0.6, 0.6, 0.4, 0.0, 0.6, 0.4, 0.0, 1.0, 0.6, 0.4, 1.0, 0.6, 0.4],
        "key_bodies": ["right_hand", "left_hand", "right_foot", "left_foot"],
        "key_pos_w": 0.15,
        "max_obs_h": 3.0,
        "mgdm": {"empty": 0},
        "min_obs_h": -3.0,
        "plane": {"dynamic_friction": 1.0, "restitution": 0.0, "static_friction": 1.0},
        "pose_termination": True,
        "pose_termination_dist": [0.7, 1.0, 0.7, 0.7, 0.7, 0.7, 0.7, 0.7, 1.0, 1.2, 10.0, 1.0, 1.2, 10.0],
        "pose_w": 0.5,
        "rand_reset": True,
        "rand_root_pos_offset_scale": 0.075,
        "ray_angle": 0.26179938779,
        "ray_dx": 0.05,
        "ray_num_left": 3,
        "ray_num_right": 3,
        "ray_points_ahead": 60,
        "ray_points_behind": 2,
        "ref_char_offset": [0.0, 0.0, 0.0],
        "rel_deepmimic_w": 1.0,
        "rel_task_w": 0.0,
        "root_height_obs": False,
        "root_pos_termination_dist": 0.6,
        "root_pos_w": 0.15,
        "root_rot_termination_angle": 1.309,
        "root_vel_w": 0.1,
        "sim_freq": 60,
        "start_paused": False,
        "tar_obs_steps": [1, 2, 3, 10, 20, 30],
        "target_motion_height_buffer": 0.01,
        "target_radius": 1.0,
        "task1_w": 0.7,
        "task2_w": 0.3,
        "termination_height": 0.15,
        "track_root": True,
        "track_root_h": True,
        "use_contact_info": True,
        "vel_w": 0.1,
        "write_agent_states": False,
    },
    "sim": {
        "physx": {"bounce_threshold_velocity": 0.2, "contact_offset": 0.02, "default_buffer_size_multiplier": 10.0,
                  "max_depenetration_velocity": 10.0, "num_position_iterations": 4, "num_threads": 4,
                  "num_velocity_iterations": 0, "rest_offset": 0.0, "solver_type": 1},
        "substeps": 2,
    },
}

_AGENT = {
    "agent_name": "DM_PPO",
    "action_bound_weight": 10.0, "action_entropy_weight": 0.0, "action_reg_weight": 0.0,
    "batch_size": 4, "clip_grad_norm": True, "critic_loss_type": "L2", "critic_loss_weight": 10.0,
    "debug_log": "output/debug_log.txt", "discount": 0.99, "exp_name": "parkour_dataset_exp001",
    "iters_per_checkpoint": 200, "iters_per_output": 100, "max_grad_norm": 1000.0,
    "model": {"action_std": 0.05, "actor_init_output_scale": 0.01, "actor_net": "fc_3layers_2048units",
              "actor_std_type": "FIXED", "critic_net": "fc_3layers_2048units"},
    "norm_adv_clip": 4.0, "norm_obs_clip": 10.0, "normalizer_samples": 300000000,
    "optimizer": {"learning_rate": 5e-5, "momentum": 0.9, "type": "SGD", "weight_decay": 0.0},
    "ppo_clip_ratio": 0.2, "project_name": "parkour", "steps_per_iter": 32, "td_lambda": 0.95,
    "test_episodes": 16, "update_epochs": 5, "use_wandb": False,
}


def default_env_config(char_file=None, motion_file=None, terrain_save_path=None):
    cfg = copy.deepcopy(_ENV)
    if char_file is None:
        from ...assets import humanoid_spec
        char_file = humanoid_spec.write_mjcf()
    cfg["env"]["char_file"] = char_file
    cfg["env"]["dm"]["motion_file"] = motion_file
    cfg["env"]["dm"]["terrain_save_path"] = terrain_save_path
    return cfg


def default_agent_config():
    return copy.deepcopy(_AGENT)
