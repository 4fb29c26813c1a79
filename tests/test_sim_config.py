"""The simulator's CONFIGURATION against fixture G23 (CPU).

Row a1 of SURVEY section 8 replaces Isaac Gym; its arithmetic is an absent binary (parity unpinned), so what CAN be pinned is that the
HIP simulator is configured from exactly the numbers the reference hands to Isaac Gym.  tests/golden/g23_sim_config.json holds them as
the reference's own files and set-up code produce them (gen_golden.py stage sim-config): data/assets/humanoid.xml:22-139 parsed with
ElementTree, envs/ig_env.py:131-164 `_parse_sim_params` + envs/ig_char_env.py:105-146 run on recording stand-ins of the gymapi objects,
PARC/tracker_config/dm_env_default.yaml + dm_agent_default.yaml.  One edit to parc_amd/assets/humanoid_spec.py, sim_model.py or
default_config.py that moves a gain, a range, a geom or a solver setting fails here.
"""
import json
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def g23():
    with open(os.path.join(HERE, "golden", "g23_sim_config.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def model():
    from parc_amd.anim.kin_char_model import KinCharModel
    from parc_amd.assets import humanoid_spec
    from parc_amd.sim_model import SimModel
    km = KinCharModel("cpu")
    km.load_char_file(humanoid_spec.write_mjcf())
    return km, SimModel(km)


def test_drives_limits_and_torque_limits_equal_the_reference_mjcf(g23, model):
    """per dof: kp = stiffness, kd = damping, armature, range (degrees in the MJCF, radians in the model), torque limit = motor gear"""
    km, sm = model
    s = sm.struct
    bodies = g23["mjcf"]["bodies"]
    assert [b["name"] for b in bodies] == list(km.get_body_names())                     # depth-first document order (_check_char_model)
    assert [b["parent"] for b in bodies] == [None if p < 0 else km.get_body_name(int(p)) for p in km._parent_indices.tolist()]
    gear = {m["joint"]: m["gear"] for m in g23["mjcf"]["motors"]}
    assert len(gear) == 28 == s.dof_size
    d = 0
    for b, spec in enumerate(bodies):
        np.testing.assert_allclose(list(s.local_translation[b]), spec["pos"], rtol=0, atol=1e-7)
        jt = km._joints[b]
        assert jt.get_dof_dim() == len(spec["joints"]), spec["name"]                    # 3 hinges = one spherical joint, 1 = hinge, 0 = fixed
        if len(spec["joints"]) > 0:
            assert jt.dof_idx == d
        if len(spec["joints"]) == 1:
            np.testing.assert_allclose(list(s.joint_axis[b]), spec["joints"][0]["axis"], atol=0)
        if len(spec["joints"]) == 3:
            assert [j["axis"] for j in spec["joints"]] == [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]      # dof k = exp-map component k
        for j in spec["joints"]:
            assert j["type"] == "hinge" and j["limited"] == "true"
            assert s.kp[d] == np.float32(j["stiffness"]) and s.kd[d] == np.float32(j["damping"]) and s.armature[d] == np.float32(j["armature"]), j["name"]
            assert abs(s.limit_lo[d] - math.radians(j["range_deg"][0])) < 1e-6 and abs(s.limit_hi[d] - math.radians(j["range_deg"][1])) < 1e-6, j["name"]
            assert s.effort[d] == np.float32(gear[j["name"]]), j["name"]
            d += 1
    assert d == 28
    # the per-joint values override the class defaults everywhere (stiffness 5 / damping 0.1 / armature 0.007 never reach a dof)
    dj = g23["mjcf"]["defaults"]["body"]["joint"]
    assert all(s.kp[k] != np.float32(float(dj["stiffness"])) for k in range(28))


def test_collision_geometry_and_mass_equal_the_reference_mjcf(g23, model):
    from parc_amd.anim.kin_char_model import GeomType
    from parc_amd.sim_model import geom_mass_properties
    km, sm = model
    total = 0.0
    for b, spec in enumerate(g23["mjcf"]["bodies"]):
        geoms = km.get_geoms(b)
        assert len(geoms) == len(spec["geoms"]), spec["name"]
        for g, e in zip(geoms, spec["geoms"]):
            assert g._density == e["density"]
            if e["type"] == "sphere":
                assert g._shape_type == GeomType.SPHERE
                np.testing.assert_allclose(g._offset, e.get("pos", [0.0, 0.0, 0.0]), atol=1e-12)
                assert float(np.atleast_1d(g._dims)[0]) == e["size"][0]
                vol = 4.0 / 3.0 * math.pi * e["size"][0] ** 3
            elif e["type"] == "capsule":
                assert g._shape_type == GeomType.CAPSULE and float(g._radius) == e["size"][0]
                np.testing.assert_allclose(g._offset, e["fromto"][0:3], atol=1e-12)
                np.testing.assert_allclose(g._offset + g._dims, e["fromto"][3:6], atol=1e-12)
                L = float(np.linalg.norm(np.array(e["fromto"][3:6]) - np.array(e["fromto"][0:3])))
                vol = math.pi * e["size"][0] ** 2 * L + 4.0 / 3.0 * math.pi * e["size"][0] ** 3
            else:
                assert e["type"] == "box" and g._shape_type == GeomType.BOX
                np.testing.assert_allclose(g._offset, e["pos"], atol=1e-12)
                np.testing.assert_allclose(g._dims, e["size"], atol=1e-12)              # MJCF box size = half extents
                vol = 8.0 * e["size"][0] * e["size"][1] * e["size"][2]
            m = geom_mass_properties(g)[0]
            assert abs(m - e["density"] * vol) < 1e-9 * max(1.0, m)                      # mass = density x volume of the reference's geom
            total += e["density"] * vol
            assert e["friction"][0] == 1.0 and e["condim"] == 1
    assert abs(sm.total_mass - total) < 1e-6 and abs(float(sum(sm.struct.mass[b] for b in range(15))) - total) < 1e-3
    assert 49.0 < total < 51.0


def test_solver_settings_equal_what_the_reference_hands_to_isaac_gym(g23, model):
    """envs/ig_env.py:131-164 + the YAML's sim block, envs/ig_char_env.py:105-146, util/ig_util.py:6-22 -> the step the env issues"""
    from parc_amd.envs.ig_parkour.default_config import default_env_config
    km, sm = model
    ig = g23["isaac_gym"]
    sp = ig["sim_params"]
    cfg = default_env_config()
    # timing: sim_freq 60 / control_freq 30 -> 2 simulate() calls per env step, each `substeps` = 2 PhysX sub-steps: 4 steps of 1/120 s
    env_cfg = cfg["env"]
    sim_steps = int(env_cfg["sim_freq"] / env_cfg["control_freq"])
    assert sim_steps == ig["sim_steps_per_control_step"] == 2 and abs(1.0 / env_cfg["sim_freq"] - sp["dt"]) < 1e-15
    assert cfg["sim"]["substeps"] == sp["substeps"] == 2
    assert abs(1.0 / (env_cfg["sim_freq"] * cfg["sim"]["substeps"]) - sp["dt"] / sp["substeps"]) < 1e-15
    assert abs(1.0 / env_cfg["control_freq"] - ig["control_dt"]) < 1e-15
    # the formulas IGParkourEnv.__init__ uses (ig_parkour_env.py:70-77) are these: keep them in step with the source
    import inspect
    from parc_amd.envs.ig_parkour import ig_parkour_env
    src = inspect.getsource(ig_parkour_env.IGParkourEnv.__init__)
    assert 'self._sim_steps = int(sim_freq / control_freq)' in src and 'self._sim_h = 1.0 / (sim_freq * self._substeps)' in src
    # gravity: z-up, -9.81
    assert sp["up_axis"] == "UP_AXIS_Z" and sp["gravity"] == {"x": 0, "y": 0, "z": -9.81} and sm.struct.gravity == np.float32(9.81)
    # every PhysX setting of the YAML is carried in the default config (consumed or not, DESIGN section 3 says which)
    for k, v in sp["physx"].items():
        if k in ("max_gpu_contact_pairs", "num_subscenes", "use_gpu"):
            continue                                                                   # set in code (ig_env.py:144-153), not in the YAML
        assert cfg["sim"]["physx"][k] == v, k
    # asset options and actor creation
    ao = ig["asset_options"]
    assert sm.struct.angular_damping == np.float32(ao["angular_damping"]) and sm.struct.max_angular_velocity == np.float32(ao["max_angular_velocity"])
    assert ao["default_dof_drive_mode"] == "DOF_MODE_POS" and env_cfg["control_mode"] == "pd"
    calls = {list(c.keys())[0]: list(c.values())[0] for c in ig["calls"]}
    assert calls["create_actor"]["collision_group"] == 7 and calls["create_actor"]["collision_filter"] == 0
    # collision group = env id: envs never touch each other; filter 0: links of one character DO collide -> every non-adjacent pair is on
    par = km._parent_indices.tolist()
    for b in range(15):
        want = sum(1 << j for j in range(15) if j != b and par[b] != j and par[j] != b)
        assert sm.struct.self_mask[b] == want
    assert calls["set_actor_dof_properties"] == {"driveMode": "DOF_MODE_POS", "stiffness": "from_asset", "damping": "from_asset"}   # pd: gains untouched
    # terrain mesh material: friction 1 / 1, restitution 0 -> one friction coefficient, no restitution term in the contact model
    tm = calls["add_triangle_mesh"]
    assert tm["static_friction"] == tm["dynamic_friction"] == 1.0 == float(sm.struct.friction_mu) and tm["restitution"] == 0.0
    assert env_cfg["plane"] == {"dynamic_friction": 1.0, "restitution": 0.0, "static_friction": 1.0}


def _diff(a, b, path=""):
    out = []
    if isinstance(a, dict) and isinstance(b, dict):
        for k in sorted(set(a) | set(b)):
            if k not in a or k not in b:
                out.append(path + "/" + k)
            else:
                out.extend(_diff(a[k], b[k], path + "/" + k))
    elif a != b:
        out.append(path)
    return out


def test_default_configs_equal_the_reference_yaml(g23):
    """default_env_config() / default_agent_config() = PARC/tracker_config/dm_env_default.yaml / dm_agent_default.yaml, key by key, except
    the paths (character file, dataset file, terrain cache, the dataset's class names) and the logger switch"""
    from parc_amd.envs.ig_parkour.default_config import default_agent_config, default_env_config
    env_diff = _diff(default_env_config(), g23["env_yaml"])
    assert env_diff == ["/env/char_file", "/env/dm/motion_classes", "/env/dm/motion_file", "/env/dm/terrain_save_path"], env_diff
    assert g23["env_yaml"]["env"]["dm"]["has_motion_classes"] is False                  # the class names are unused by default
    agent = default_agent_config()
    agent_diff = _diff(agent, g23["agent_yaml"])
    assert agent_diff == ["/optimizer/learning_rate", "/use_wandb"], agent_diff
    # PyYAML reads `5e-5` (no dot) as a string; the value is the same
    assert agent["optimizer"]["learning_rate"] == float(g23["agent_yaml"]["optimizer"]["learning_rate"]) == 5e-5
    assert agent["use_wandb"] is False and g23["agent_yaml"]["use_wandb"] is True       # wandb is not installable here
