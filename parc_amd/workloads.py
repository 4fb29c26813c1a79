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
}


def build_env(name, num_envs, device, seed=0):
    spec = WORKLOADS[name]
    clips = synthetic.make_dataset(num_clips=spec["num_clips"], seed=seed, tile_cells=spec["tile_cells"], frames_range=spec["frames_range"],
                                   flat=spec["flat"], terrain_kind=spec.get("terrain_kind", "boxes"))
    tiled = synthetic.tile_square(clips)
    cfg = default_env_config()
    return IGParkourEnv(cfg, num_envs, device, False, motion_input=clips, tiled_terrain=tiled), clips, tiled


def build_agent(env, device, **overrides):
    from .learning.dm_ppo_agent import DMPPOAgent
    cfg = default_agent_config()
    cfg.update(overrides)
    return DMPPOAgent(cfg, env, device)
