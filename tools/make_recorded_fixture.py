"""Produce the files the tracker writes for the reference's readers -- a recorded clip (parc_4_phys_record) and the
terrain.pkl cache -- on the GPU, small enough to commit under tests/golden/recorded/.  tests/golden/gen_golden.py then
opens them with the REFERENCE's own MotionLib / load_terrain in the build container and stores what it read as fixture
G19, which the tests compare with this package's view of the same files.

    gpurun -- python tools/make_recorded_fixture.py gpurun_out/recorded       # then copy into tests/golden/recorded/
"""
import os
import shutil
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir):
    import parc_amd
    from parc_amd import synthetic
    from parc_amd.assets import humanoid_spec
    from parc_amd.envs import env_builder
    from parc_amd.envs.ig_parkour.default_config import default_agent_config, default_env_config
    from parc_amd.learning import agent_builder
    from parc_amd.util import terrain_util
    parc_amd.install_reference_aliases()
    os.makedirs(out_dir, exist_ok=True)
    tmp = os.path.join(out_dir, "_work")
    os.makedirs(tmp, exist_ok=True)
    torch.manual_seed(0)
    clips = synthetic.make_dataset(2, seed=5, tile_cells=12, frames_range=(40, 50), boxes=4)
    entries = []
    for c in clips:
        ter = terrain_util.SubTerrain.from_arrays(c["hf"], c["min_point"], c["dxdy"], device="cpu").numpy_copy()
        p = os.path.join(tmp, c["name"] + ".pkl")
        terrain_util.dump_reference_pickle({"fps": 30, "loop_mode": "CLAMP", "frames": c["frames"], "contacts": c["contacts"],
                                            "terrain": ter}, p)
        entries.append({"file": p, "weight": 1.0})
    motions = os.path.join(tmp, "motions.yaml")
    with open(motions, "w") as f:
        yaml.safe_dump({"motions": entries}, f)
    cache = os.path.join(tmp, "terrain.pkl")
    env_cfg = default_env_config(char_file=humanoid_spec.write_mjcf(), motion_file=motions, terrain_save_path=cache)
    env_cfg["env"]["output_motion_dir"] = os.path.join(tmp, "recorded")
    agent_cfg = default_agent_config()
    env_yaml, agent_yaml = os.path.join(tmp, "dm_env.yaml"), os.path.join(tmp, "agent_config.yaml")
    with open(env_yaml, "w") as f:
        yaml.safe_dump(env_cfg, f)
    with open(agent_yaml, "w") as f:
        yaml.safe_dump(agent_cfg, f)
    env = env_builder.build_env(env_yaml, 2, "cuda:0", False)
    agent = agent_builder.build_agent(agent_yaml, env, "cuda:0")
    env._bypass_record_fail = True          # an untrained policy falls early: the writer is what is exercised
    agent.record_motions(max_steps=30)
    files = sorted(os.listdir(env._output_motion_dir))
    assert files, "nothing recorded"
    shutil.copy(os.path.join(env._output_motion_dir, files[0]), os.path.join(out_dir, "recorded_clip_dm.pkl"))
    shutil.copy(cache, os.path.join(out_dir, "terrain.pkl"))
    shutil.rmtree(tmp)
    for f in sorted(os.listdir(out_dir)):
        print(f, os.path.getsize(os.path.join(out_dir, f)), "bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/recorded")
