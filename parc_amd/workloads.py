"""Named synthetic workloads = the BASELINE.json configs that fit one GPU (used by bench.py, smoke and tests)."""
from . import synthetic
from .envs.ig_parkour.default_config import default_agent_config, default_env_config
from .envs.ig_parkour.ig_parkour_env import IGParkourEnv

WORKLOADS = {
    # BASELINE.json configs[1]: 1024-env tracker, flat terrain, single reference clip
    "flat_1clip": dict(num_clips=1, flat=True, tile_cells=50, frames_range=(58, 58)),
    # BASELINE.json configs[2]: 4096-env tracker on procgen box heightfields (64 clips, 16x16 @ 0.4 m tiles)
    "boxes_64clips": dict(num_clips=64, flat=False, tile_cells=16, frames_range=(120, 254)),
    # BASELINE.json configs[4] terrains: stairs / curvy paths / both / boxes from the reference's generators, 32x32 @ 0.4 m tiles
    "parkour_32clips": dict(num_clips=32, flat=False, tile_cells=32, frames_range=(120, 200), terrain_kind="parkour"),
    # BASELINE.json configs[3] stand-in for the full iter-0 dataset (not downloadable): 1024 clips of 2-10 s at
    # 30 fps, terrains of 16..45 cells per side @ 0.4 m -> 32 x 32 tiles of <= 47^2 cells = a <= 1504^2 heightfield (~9 MB) and ~80 MB
    # of clip rows: the one workload whose clip database and heightfield do NOT fit in L2
    "iter0_1024clips": dict(num_clips=1024, flat=False, tile_cells=45, tile_cells_range=(16, 45), frames_range=(61, 301)),
}


def shipped_clip(which=0):
    """One of the two real clips the reference ships (data/terrains/civilization.pkl = 0, TEASER_TERRAIN.pkl = 1) with its own
    terrain, read from the committed fixtures tests/golden/g3_motion.npz + g5_hf_*.npz (the reference itself is not on the GPU box)."""
    import os
    import numpy as np
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    z3 = np.load(os.path.join(gold, "g3_motion.npz"))
    z5 = np.load(os.path.join(gold, ("g5_hf_civ", "g5_hf_teaser")[which] + ".npz"))
    return dict(frames=z3["frames_%d" % which].astype(np.float32), contacts=z3["contacts_%d" % which].astype(np.float32), fps=30.0, loop=0,
                weight=1.0, hf=z5["hf"].astype(np.float32), min_point=z5["min_point"].astype(np.float32), dxdy=z5["dxdy"].astype(np.float32),
                name=("civilization", "teaser")[which])


def build_env(name, num_envs, device, seed=0, env_overrides=None):
    """-> (env, clips, tiled).  env_overrides: keys of the env YAML's `env:` block to replace (e.g. fraction_dm_envs + mgdm)."""
    if name in ("civ_clip", "teaser_clip"):          # the authors' own clip on its own terrain (learning evidence on real data)
        clips = [shipped_clip(0 if name == "civ_clip" else 1)]
        tiled = synthetic.tile_square(clips)
        return IGParkourEnv(default_env_config(), num_envs, device, False, motion_input=clips, tiled_terrain=tiled), clips, tiled
    spec = WORKLOADS[name]
    clips = synthetic.make_dataset(num_clips=spec["num_clips"], seed=seed, tile_cells=spec["tile_cells"], frames_range=spec["frames_range"],
                                   flat=spec["flat"], terrain_kind=spec.get("terrain_kind", "boxes"),
                                   tile_cells_range=spec.get("tile_cells_range"))
    tiled = synthetic.tile_square(clips)
    cfg = default_env_config()
    if env_overrides:
        cfg["env"].update(env_overrides)
    return IGParkourEnv(cfg, num_envs, device, False, motion_input=clips, tiled_terrain=tiled), clips, tiled


def build_agent(env, device, **overrides):
    from .learning.dm_ppo_agent import DMPPOAgent
    cfg = default_agent_config()
    cfg.update(overrides)
    return DMPPOAgent(cfg, env, device)


def build_core(name, num_envs, device, seed=0):
    """The tracker core alone (no simulator, no agent) on a named workload, every env placed ON its reference pose at a random clip
    time: the state the per-kernel benchmarks and profiles launch the fused post-step kernel on.  -> (core, clips, tiled)"""
    import torch
    from . import _hip
    from .anim.kin_char_model import KinCharModel
    from .anim.motion_lib import MotionLib
    from .assets import humanoid_spec
    from .tracker_core import TrackerConfig, TrackerCore
    from .util import geom_util
    from .util.terrain_util import SubTerrain
    spec = WORKLOADS[name]
    km = KinCharModel(device)
    km.load_char_file(humanoid_spec.write_mjcf())
    clips = synthetic.make_dataset(num_clips=spec["num_clips"], seed=seed, tile_cells=spec["tile_cells"], frames_range=spec["frames_range"],
                                   flat=spec["flat"], terrain_kind=spec.get("terrain_kind", "boxes"), tile_cells_range=spec.get("tile_cells_range"))
    M = len(clips)
    mlib = MotionLib(clips, km, device, init_type="clips", contact_info=True)
    tiled = synthetic.tile_square(clips)
    hf, mn, dxdy, offs = tiled
    rays = geom_util.get_xy_points_cone(torch.zeros(2), 0.05, 2, 60, 3, 3, 0.26179938779)
    cfg = TrackerConfig(default_env_config()["env"], km, rays.shape[0])
    core = TrackerCore(num_envs, device, km, mlib, cfg, rays)
    core.set_terrain(SubTerrain.from_arrays(hf, mn, dxdy, device=device))
    g = torch.Generator().manual_seed(seed)
    n = num_envs
    core.motion_ids[:] = torch.randint(0, M, (n,), generator=g).to(device)
    core.motion_xy_offset[:] = torch.tensor(offs[:, 0]).to(device)[core.motion_ids]
    lens = mlib._motion_lengths[core.motion_ids]
    core.motion_time_offsets[:] = torch.rand(n, generator=g).to(device) * (lens - 1.9).clamp_min(0.0)
    core.time_buf[:] = (torch.randint(1, 55, (n,), generator=g).float() / 30.0).to(device)
    core.post_step(_hip.POST_REF)
    core.root_state[:, 0:3] = core.ref_root_pos
    core.root_state[:, 3:7] = core.ref_root_rot
    core.dof_state.view(n, 28, 2)[..., 0] = core.ref_dof_pos
    core.rigid_body_state.view(n, 15, 13)[..., 0:3] = core.ref_body_pos
    return core, clips, tiled


def eager_rollout_like_the_graph(agent, num_steps):
    """Measurement helper (bench.py, tools/rollout_only.py --eager): the launches the captured rollout step consists of - same
    kernels, same order, same device-side restart of finished envs, device-scalar write row - issued eagerly, so that a single launch can
    be bracketed by events or read from a profiler trace (kernel nodes of a replayed hipGraph can be neither: events captured into a
    graph are not re-recorded by a replay, and rocprofv3 inflates the duration of every graph node by ~3 us)."""
    env, eb = agent._env, agent._exp_buffer
    snapshots = getattr(env, "_info_snapshots", None)
    if snapshots is not None:
        env._info_snapshots = False
    eb.set_device_head(agent._head_t)
    agent._in_graph_step = True
    try:
        device_reset = hasattr(env, "reset_done") and env.supports_device_reset()
        for _ in range(num_steps):
            # (with the step's own tick - DMPPOAgent._device_tick - the cell holds the row of the step before)
            agent._head_t.fill_((eb._buffer_head - 1) % eb._buffer_length if agent._device_tick() else eb._buffer_head)
            agent._exp_prob_t.fill_(agent._get_exp_prob())
            done = agent._train_step_body(device_reset)
            if not device_reset:
                agent._curr_obs, agent._curr_info = agent._reset_done_envs(done)
            eb.inc()
    finally:
        eb.set_device_head(None)
        agent._head_dev = -1                 # the device's write row was moved behind the agent's back: re-written before the next replay
        agent._in_graph_step = False
        if snapshots is not None:
            env._info_snapshots = snapshots
