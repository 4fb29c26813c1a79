"""Stage 2's post-processing chain feeding the tracker, end to end on the GPU, every hand-off through files in the reference's format
(parc_2_kin_gen.py:427-511 -> PARC/util/create_dataset.py -> parc_3_tracker's env): optimise a generated clip on its terrain, drop
hesitation frames, compute the per-frame heightfield masks, save it and its mirrored copy, build the class-balanced dataset YAML with
preprocessing, construct the tracking env from that YAML, step it."""
import numpy as np
import pytest
import torch

from test_hip_parity import DEV, km  # noqa: F401  (km is a fixture)

pytestmark = pytest.mark.gpu


def test_stage2_outputs_feed_the_tracker(km, tmp_path):
    from parc_amd import synthetic
    from parc_amd.assets import humanoid_spec
    from parc_amd.envs.ig_parkour.default_config import default_env_config
    from parc_amd.envs.ig_parkour.ig_parkour_env import IGParkourEnv
    from parc_amd.tools.motion_opt import motion_optimization as mo
    from parc_amd.util import create_dataset, geom_util, safe_pickle, terrain_util, torch_util
    from parc_amd.zmotion_editing_tools import motion_edit_lib as medit
    out_dir = tmp_path / "kin_gen" / "boxes"
    out_dir.mkdir(parents=True)
    body_points = geom_util.get_char_point_samples(km)
    w = dict(w_root_pos=1.0, w_root_rot=10.0, w_joint_rot=1.0, w_smoothness=10.0, w_penetration=1000.0, w_contact=1000.0, w_sliding=10.0,
             w_body_constraints=1000.0, w_jerk=1000.0)
    clips = synthetic.make_dataset(num_clips=2, seed=11, frames_range=(70, 90), tile_cells=16)
    names = []
    for k, clip in enumerate(clips):
        frames = torch.tensor(clip["frames"], dtype=torch.float32, device=DEV)
        frames[:, 2] -= 0.03                                        # a generated clip that scrapes the ground
        frames[30:36] = frames[29]                                  # ... and dithers for a few frames
        contacts = torch.tensor(clip["contacts"], dtype=torch.float32, device=DEV)
        ter = terrain_util.SubTerrain.from_arrays(clip["hf"], clip["min_point"], clip["dxdy"], device=DEV)
        bc = mo.compute_approx_body_constraints(frames[:, 0:3].contiguous(), torch_util.exp_map_to_quat(frames[:, 3:6]),
                                                km.dof_to_rot(frames[:, 6:].contiguous()), contacts, km, ter)
        trace = []
        opt = mo.motion_contact_optimization(src_frames=frames, contacts=contacts, body_points=body_points, terrain=ter, char_model=km, num_iters=60,
                                             step_size=0.001, body_constraints=bc, max_jerk=1000.0, exp_name="t", use_wandb=False, log_file=None,
                                             verbose=False, loss_trace=trace, **w)
        assert float(trace[0][-1]) < float(trace[0][0])
        opt, con = medit.remove_hesitation_frames(opt.cpu(), contacts.cpu(), km)          # host tensors, as parc_2_kin_gen.py:469 passes them
        assert opt.shape[0] <= frames.shape[0] - 4 and opt.device.type == "cpu"
        opt = opt.to(DEV)
        inds = terrain_util.compute_hf_extra_vals(motion_frames=opt, terrain=ter, char_model=km, char_body_points=body_points)
        name = "BOXES_%d_0_opt" % k
        medit.save_motion_data(str(out_dir / (name + ".pkl")), opt, con, ter, 30, "CLAMP", loss=0.5, hf_mask_inds=inds,
                               **{"opt:body_constraints": [[_cpu(c) for c in lst] for lst in bc]})
        fl, flc = medit.flip_motion_about_XZ_plane(motion_frames=opt, char_model=km, contact_frames=con)
        fter = ter.torch_copy()
        fter.flip_by_XZ_axis()
        medit.save_motion_data(str(out_dir / (name + "_flipped.pkl")), fl, flc, fter, 30, "CLAMP", loss=0.5)
        names += [name, name + "_flipped"]
    ds = tmp_path / "motions.yaml"
    entries = create_dataset.create_dataset_yaml([tmp_path / "kin_gen"], ds, char_filepath=humanoid_spec.write_mjcf(), compute_preprocessing_data=True)
    assert len(entries) == 4 and abs(sum(e["weight"] for e in entries) - sum(_len(e["file"]) for e in entries)) < 1e-3
    back = safe_pickle.load_motion_file_safe(str(out_dir / (names[0] + ".pkl")))
    assert "hf_mask_inds" in back and "opt:body_constraints" in back and back["terrain"]["hf"].shape == (16, 16)
    # the tracker on that dataset
    cfg = default_env_config(char_file=humanoid_spec.write_mjcf(), motion_file=str(ds), terrain_save_path=str(tmp_path / "terrain.pkl"))
    env = IGParkourEnv(cfg, 64, DEV, False)
    dm = env.get_dm_env()
    assert dm._motion_lib.num_motions() == 4 and sorted(dm._motion_lib.get_motion_names()) == sorted(names)
    obs, _ = env.reset()
    assert torch.isfinite(obs).all()
    for _ in range(5):
        obs, r, done, info = env.step(torch.clamp(env._ref_dof_pos, env._action_bound_low, env._action_bound_high))
    assert torch.isfinite(obs).all() and torch.isfinite(r).all() and float(r.mean()) > 0.2
    # a mirrored clip is the mirror image of its source: root y of frame 0 flips sign
    ml = dm._motion_lib
    i0, i1 = ml.get_motion_names().index(names[0]), ml.get_motion_names().index(names[1])
    f0 = ml._motion_frames[ml._motion_start_idx[i0]]
    f1 = ml._motion_frames[ml._motion_start_idx[i1]]
    assert abs(float(f0[1] + f1[1])) < 1e-6 and abs(float(f0[0] - f1[0])) < 1e-6


def _cpu(c):
    c.constraint_point = c.constraint_point.cpu()
    return c


def _len(path):
    from parc_amd.util import safe_pickle
    d = safe_pickle.load_motion_file_safe(path)
    return np.asarray(d["frames"]).shape[0] / float(d["fps"])              # the dataset builder counts frames, not intervals


def test_optimize_motions_driver_from_a_config_file(km, tmp_path):
    """tools/motion_opt/optimize_motions.py as a program: the reference's config keys, a dataset YAML in, <name>_opt.pkl files in the
    reference's motion format out (frame_stride 2 halves the clip and its fps, constraints are re-indexed)."""
    import yaml
    from parc_amd import synthetic
    from parc_amd.assets import humanoid_spec
    from parc_amd.tools.motion_opt import optimize_motions
    from parc_amd.util import safe_pickle, terrain_util
    from test_host_logic import pickle_globals
    src = tmp_path / "src"
    src.mkdir()
    entries = []
    for k, clip in enumerate(synthetic.make_dataset(num_clips=2, seed=21, frames_range=(60, 70), tile_cells=16)):
        ter = terrain_util.SubTerrain.from_arrays(clip["hf"], clip["min_point"], clip["dxdy"], device="cpu").numpy_copy()
        path = str(src / ("clip_%d.pkl" % k))
        terrain_util.dump_reference_pickle({"fps": 30, "loop_mode": "CLAMP", "frames": clip["frames"], "contacts": clip["contacts"], "terrain": ter}, path)
        entries.append({"file": path, "weight": 1.0})
    (tmp_path / "motions.yaml").write_text(yaml.safe_dump({"motions": entries}))
    cfg = {"motions_yaml_path": str(tmp_path / "motions.yaml"), "device": DEV, "char_model": humanoid_spec.write_mjcf(),
           "output_folder_path": str(tmp_path / "opt") + "/", "num_iters": 30, "step_size": 0.001, "w_root_pos": 1.0, "w_root_rot": 10.0,
           "w_joint_rot": 1.0, "w_smoothness": 10.0, "w_penetration": 1000.0, "w_contact": 1000.0, "w_sliding": 10.0, "w_body_constraints": 1000.0,
           "w_jerk": 1000.0, "max_jerk": 1000.0, "use_wandb": False, "auto_compute_body_constraints": True, "frame_stride": 2,
           "char_point_samples": {"sphere_num_subdivisions": 0, "box_num_slices": 2, "box_dim_x": 3, "box_dim_y": 6, "capsule_num_circle_points": 4,
                                  "capsule_num_sphere_subdivisions": 0, "capsule_num_cylinder_slices": 4}}
    (tmp_path / "motion_opt.yaml").write_text(yaml.safe_dump(cfg))
    optimize_motions.main(["optimize_motions.py", "--config", str(tmp_path / "motion_opt.yaml")])
    for k in range(2):
        out = tmp_path / "opt" / ("clip_%d_opt.pkl" % k)
        assert out.exists() and (tmp_path / "opt" / "log" / ("log_clip_%d_opt.txt" % k)).exists()
        d = safe_pickle.load_motion_file_safe(str(out))
        n_src = safe_pickle.load_motion_file_safe(entries[k]["file"])["frames"].shape[0]
        assert d["fps"] == 15 and d["frames"].shape == ((n_src + 1) // 2, 34) and d["contacts"].shape[0] == d["frames"].shape[0]
        assert np.isfinite(np.asarray(d["frames"])).all()
        mods = {m for m, _ in pickle_globals(str(out))}
        assert not any(m.startswith("parc_amd") for m in mods), mods
