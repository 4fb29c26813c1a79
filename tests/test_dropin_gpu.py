"""End-to-end drop-in path on the GPU, driven exactly like the reference's stage scripts drive run.py
(parc_3_tracker.py:8-78 writes env/agent YAML and calls run.main; parc_4_phys_record.py:8-65 runs --mode record):
motion pickles + dataset YAML on disk -> env_builder / agent_builder from YAML files -> train a few iterations ->
checkpoint -> reload -> record mode writes motion files in the PARC motion format."""
import os
import pickle

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def _write_dataset(tmp, n_clips=4):
    import parc_amd
    from parc_amd import synthetic
    from parc_amd.util.terrain_util import SubTerrain
    parc_amd.install_reference_aliases()
    clips = synthetic.make_dataset(n_clips, seed=3, frames_range=(50, 70), flat=True)
    entries = []
    for c in clips:
        ter = SubTerrain.from_arrays(c["hf"], c["min_point"], c["dxdy"], device="cpu").numpy_copy()
        p = os.path.join(tmp, c["name"] + ".pkl")
        with open(p, "wb") as f:
            pickle.dump({"fps": 30, "loop_mode": "CLAMP", "frames": c["frames"], "contacts": c["contacts"], "terrain": ter}, f)
        entries.append({"file": p, "weight": 1.0})
    ypath = os.path.join(tmp, "motions.yaml")
    with open(ypath, "w") as f:
        yaml.safe_dump({"motions": entries}, f)
    return ypath


def test_train_checkpoint_record_through_run_main(tmp_path):
    import parc_amd
    from parc_amd import run as parc_run
    from parc_amd.assets import humanoid_spec
    from parc_amd.envs.ig_parkour.default_config import default_agent_config, default_env_config
    from parc_amd.util import safe_pickle
    tmp = str(tmp_path)
    motions = _write_dataset(tmp)
    env_cfg = default_env_config(char_file=humanoid_spec.write_mjcf(), motion_file=motions, terrain_save_path=os.path.join(tmp, "terrain.pkl"))
    env_cfg["env"]["output_motion_dir"] = os.path.join(tmp, "recorded")
    agent_cfg = default_agent_config()
    agent_cfg.update(steps_per_iter=8, update_epochs=1, iters_per_output=1000, iters_per_checkpoint=1)
    env_yaml, agent_yaml = os.path.join(tmp, "dm_env.yaml"), os.path.join(tmp, "agent_config.yaml")
    with open(env_yaml, "w") as f:
        yaml.safe_dump(env_cfg, f)
    with open(agent_yaml, "w") as f:
        yaml.safe_dump(agent_cfg, f)
    model = os.path.join(tmp, "model.pt")
    argv = ["run.py", "--env_config", env_yaml, "--agent_config", agent_yaml, "--mode", "train", "--num_envs", "32", "--device", "cuda:0",
            "--visualize", "False", "--max_samples", str(2 * 8 * 32), "--out_model_file", model, "--int_output_dir", os.path.join(tmp, "ckpt"),
            "--log_file", os.path.join(tmp, "log.txt"), "--rand_seed", "0"]
    parc_run.main(argv)
    assert os.path.exists(model) and os.path.exists(os.path.join(tmp, "terrain.pkl")) and os.path.getsize(os.path.join(tmp, "log.txt")) > 0
    sd = torch.load(model, weights_only=True)
    assert "_model._actor_layers.0.weight" in sd and "_obs_norm._mean" in sd
    assert os.path.exists(os.path.join(tmp, "ckpt", "model_0000000000.pt")) and os.path.exists(os.path.join(tmp, "ckpt", "fail_rates_0000000000.pt"))
    # second construction loads the cached terrain (ig_parkour_env.py:600-611) and the checkpoint; record mode: one env per clip
    from parc_amd.envs import env_builder
    from parc_amd.learning import agent_builder
    env = env_builder.build_env(env_yaml, 4, "cuda:0", False)
    agent = agent_builder.build_agent(agent_yaml, env, "cuda:0")
    agent.load(model)
    env._bypass_record_fail = True            # an untrained policy falls early; still exercise the file writer
    succ = agent.record_motions(max_steps=80)
    files = sorted(os.listdir(env._output_motion_dir))
    assert len(succ) == 4 and len(files) >= 1
    rec = safe_pickle.load_motion_file_safe(os.path.join(env._output_motion_dir, files[0]))
    T = rec["frames"].shape[0]
    assert rec["fps"] == 30 and rec["loop_mode"] == "CLAMP" and rec["frames"].shape == (T, 34) and rec["contacts"].shape == (T, 15)
    assert rec["obs"].shape == (T, 1312) and rec["terrain"]["hf"].ndim == 2
    assert list(rec["obs_shapes"].keys()) == ["char_obs", "tar_obs", "tar_contacts", "char_contacts", "hf"]
    assert np.allclose(rec["frames"][0, 0:2], 0.0, atol=1e-5)          # localized on the first frame
    # both written files are in the REFERENCE's format: the only class they name is util.terrain_util.SubTerrain
    # (reference readers: anim/motion_lib.py:240 for the clip, envs/ig_parkour/dm_env.py:493-507 for the terrain cache)
    from test_host_logic import NUMPY_GLOBALS, TORCH_GLOBALS, pickle_globals
    sub = {("util.terrain_util", "SubTerrain")}
    assert pickle_globals(os.path.join(env._output_motion_dir, files[0])) - NUMPY_GLOBALS - {("collections", "OrderedDict")} == sub
    assert pickle_globals(os.path.join(tmp, "terrain.pkl")) - NUMPY_GLOBALS - TORCH_GLOBALS == sub
    # the recorded clip is itself a valid motion file: it loads back into a MotionLib
    from parc_amd.anim.motion_lib import MotionLib
    ml = MotionLib(os.path.join(env._output_motion_dir, files[0]), env._kin_char_model, "cuda:0", contact_info=True)
    assert ml.num_motions() == 1 and abs(ml._motion_lengths[0].item() - (T - 1) / 30.0) < 1e-5


def test_bench_two_ranks_from_a_plain_invocation():
    """`python bench.py --gpus 2` with no launcher around it (the driver's call): the script starts the two ranks itself and rank 0
    reports n_gpus 2.  On this one-GPU box the ranks share the device over gloo (PARC_DIST_BACKEND); the nccl path needs 2 GPUs."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, PARC_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--envs", "64", "--steps", "1", "--warmup", "1"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert abs(line["value"] - 2 * 64 * 32 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["scaling"] == "weak" and line["config"]["rollout_steps"] == 32 and line["config"]["minibatch"] == 4 * 64
    # the reference's own scaling rule (base_agent.py:179-180, ppo_agent.py:27-29): half the rollout and half the minibatch per rank at P = 2,
    # the gradient exchanged through the bucketed all-reduce on every one of the 5 x 8 optimizer steps
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--envs", "64", "--steps", "1", "--warmup", "1",
                          "--scaling", "reference"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["rollout_steps"] == 16 and line["config"]["minibatch"] == 2 * 64
    assert abs(line["value"] - 2 * 64 * 16 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    if torch.cuda.device_count() < 8:
        env.pop("PARC_DIST_BACKEND")
        bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--envs", "64", "--steps", "1"],
                             capture_output=True, text=True, env=env, timeout=300)
        assert bad.returncode != 0 and bad.stdout.strip() == ""


def test_collectives_on_the_real_backend_with_one_rank():
    """RCCL itself (torch.distributed "nccl"), world size 1, the package's multi-process switch forced on inside the probe: every
    collective of the data-parallel path (bucketed asynchronous all-reduce of gradient views started from gradient callbacks, the
    per-epoch averaging of flat parameters / momentum, broadcasts, MIN reductions) runs on the backend the 8-GPU job will use, and the
    result equals the single-process run exactly (a mean over one rank is the identity).  tools/rccl_probe.py"""
    import json
    import subprocess
    import sys
    from conftest import REPO
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PARC_DIST_BACKEND"):
        env.pop(k, None)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(REPO, "tools", "rccl_probe.py")], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    for cadence in ("minibatch", "epoch"):
        r = res[cadence]
        assert r["finite"] and r["synced"] and r["max_abs_diff"] == 0.0 and r["loss"] == r["loss_single"], (cadence, r)
    assert res["minibatch"]["overlap"] and res["minibatch"]["buckets"] >= 2
    assert res["helpers"] == {"broadcast": True, "sum": True, "min": 3.5, "mean": True}


def test_g24_files_and_argv_of_the_reference_stage_scripts_run(tmp_path, monkeypatch):
    """What the reference's own parc_3_tracker.train_tracker / parc_4_phys_record.record_motions wrote and passed to run.main when they
    ran unchanged on this package's aliases (fixture G24, tests/golden/gen_golden.py stage stage-scripts) - dm_env.yaml, agent_config.yaml,
    record_env.yaml and the three argv lists, byte for byte with the scratch directory substituted - fed to parc_amd.run.main on the
    GPU: stage 3 from scratch (writes model.pt + checkpoints + the terrain cache), stage 3 resuming from that model with the normaliser
    frozen, stage 4 in record mode with one env per dataset entry.  The dataset is the identical 4-file tree of real motion rebuilt from
    the committed fixtures; `char_file: data/assets/humanoid.xml` is relative to the launch directory like in the reference, so the
    character file is put there."""
    import json
    import sys
    from parc_amd import run as parc_run
    from parc_amd.assets import humanoid_spec
    from parc_amd.util import terrain_util
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import dataset_tree
    with open(os.path.join(here, "golden", "g24_stage_scripts.json")) as f:
        g = json.load(f)
    tmp = str(tmp_path)
    sub = lambda s: s.replace("<TMP>", tmp)

    def make_terrain(hf, min_point, dxdy):
        return terrain_util.SubTerrain.from_arrays(hf, min_point, dxdy, device="cpu").numpy_copy()
    dataset_tree.build_clip_tree(os.path.join(tmp, "tree"), make_terrain, dump=terrain_util.dump_reference_pickle)
    os.makedirs(os.path.join(tmp, "dataset"))
    with open(os.path.join(tmp, "dataset", "motions.yaml"), "w") as f:
        f.write(sub(g["dataset_yaml"]))
    os.makedirs(os.path.join(tmp, "data", "assets"))
    humanoid_spec.write_mjcf(os.path.join(tmp, "data", "assets", "humanoid.xml"))
    monkeypatch.chdir(tmp)
    for name in ("tracker_fresh", "tracker_resume", "record"):
        run = g["runs"][name]
        out_dir = sub(run["config"]["output_dir"])
        os.makedirs(out_dir, exist_ok=True)
        for fn, text in run["files_written_to_output_dir"].items():
            with open(os.path.join(out_dir, fn), "w") as f:
                f.write(sub(text))
        argv = [sub(a) for a in run["argv"][0]]
        # the arguments file the scripts leave next to the YAML is the argv (train_args.txt / record_args.txt)
        args_txt = sub(run["files_written_to_output_dir"]["record_args.txt" if name == "record" else "train_args.txt"]).split()
        assert args_txt == argv[1:]
        parc_run.main(argv)
        torch.cuda.synchronize()
        if name != "record":
            sd = torch.load(os.path.join(out_dir, "model.pt"), weights_only=True)
            assert "_model._actor_layers.0.weight" in sd and "_obs_norm._mean" in sd
            assert os.path.exists(os.path.join(out_dir, "checkpoints", "model_0000000000.pt")) and os.path.getsize(os.path.join(out_dir, "log.txt")) > 0
            assert os.path.exists(os.path.join(out_dir, "terrain.pkl"))
            if name == "tracker_resume":
                # normalizer_samples: 0 (parc_3_tracker.py:35-36): the loaded normaliser is kept as it is
                fresh = torch.load(os.path.join(tmp, "tracker_fresh", "model.pt"), weights_only=True)
                assert torch.equal(sd["_obs_norm._mean"], fresh["_obs_norm._mean"]) and torch.equal(sd["_obs_norm._count"], fresh["_obs_norm._count"])
        else:
            assert os.path.exists(os.path.join(out_dir, "terrain.pkl"))
