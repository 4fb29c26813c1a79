"""Physical invariants of the articulated-body simulator, checked on the HOST builds of the very same source
(parc_amd/csrc/parc_sim_core.h compiled by g++, oracle/sim_host.cpp; the body-per-lane kernel under a lane emulation) and - marked
`gpu` - on the DEVICE kernel the product launches (sim_step_bpl_kernel through parc_sim_step of the C ABI, tests/device_sim.py).  The
dynamics have no arithmetic reference (Isaac Gym is an absent third-party binary): parity unpinned, so correctness is argued from
conservation laws, closed forms, rest states and an independent float64 inverse dynamics - none of which shares code with the kernel."""
import copy

import numpy as np
import pytest
import torch

from conftest import REPO  # noqa: F401


_FORMULATION = ["core"]


@pytest.fixture(scope="module", params=["core", "bpl", pytest.param("device", marks=pytest.mark.gpu)], autouse=True)
def formulation(request):
    """Every invariant is checked on both formulations of the same equations: the one-env-per-lane core (parc_sim_core.h) and
    the body-per-lane kernel the product launches (parc_sim_bpl.h, run on the host by the lane emulation of sim_host_bpl.cpp) - and,
    on a GPU box, on that kernel itself as the device runs it ("device": make() then returns tests/device_sim.DeviceSim)."""
    from oracle import sim_host
    prev = sim_host.DEFAULT_VARIANT
    if request.param != "device":
        sim_host.DEFAULT_VARIANT = request.param
    _FORMULATION[0] = request.param
    yield request.param
    sim_host.DEFAULT_VARIANT = prev
    _FORMULATION[0] = "core"


def host_only():
    if _FORMULATION[0] == "device":
        pytest.skip("a statement about the host builds (runs under the 'core' / 'bpl' formulations)")


@pytest.fixture(scope="module")
def model():
    from parc_amd.anim.kin_char_model import KinCharModel
    from parc_amd.assets import humanoid_spec
    from parc_amd.sim_model import SimModel
    km = KinCharModel("cpu")
    km.load_char_file(humanoid_spec.write_mjcf())
    return km, SimModel(km)


def make(model, n=1, hf=None, **over):
    from oracle.sim_host import HostSim
    km, sm = model
    s = copy.deepcopy(sm.struct)
    if over.pop("self_collision", True) is False:
        for b in range(16):
            s.self_mask[b] = 0
    for k, v in over.items():
        setattr(s, k, v)
    if hf is None:
        hf = np.full((20, 20), -100.0, np.float32)   # ground far away: free flight
    if _FORMULATION[0] == "device":
        from device_sim import DeviceSim
        return DeviceSim(s, n, hf, [-4.0, -4.0], [0.4, 0.4], num_bodies=int(s.num_bodies), dof_size=int(s.dof_size))
    return HostSim(s, n, hf, [-4.0, -4.0], [0.4, 0.4], num_bodies=int(s.num_bodies), dof_size=int(s.dof_size))


def rotm(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def momentum(model, sim, e=0):
    """linear momentum, angular momentum about the world origin, kinetic energy from rigid_body_state"""
    _, sm = model
    P, L, T = np.zeros(3), np.zeros(3), 0.0
    for b in range(sim.B):
        bs = sim.rigid_body_state[e, b].astype(np.float64)
        R = rotm(bs[3:7])
        w, v0 = bs[10:13], bs[7:10]
        c = R @ sm.body_com[b]
        vc = v0 + np.cross(w, c)
        xc = bs[0:3] + c
        I = R @ sm.body_inertia_com[b] @ R.T
        m = sm.body_mass[b]
        P += m * vc
        L += I @ w + m * np.cross(xc, vc)
        T += 0.5 * m * vc @ vc + 0.5 * w @ I @ w
    return P, L, T


def test_mass_properties(model):
    host_only()
    _, sm = model
    assert 40.0 < sm.total_mass < 55.0          # DeepMimic humanoid: ~45 kg nominal, ~50 kg from the geom volumes
    assert np.all(sm.body_mass > 0.1)
    for b in range(15):
        assert np.all(np.linalg.eigvalsh(sm.body_inertia_com[b]) > 0)
    assert sm.struct.num_spheres == 50


def test_free_fall_closed_form(model):
    sim = make(model)
    sim.root_state[0, 2] = 5.0
    sim.root_state[0, 7:10] = [0.3, -0.2, 1.0]
    h, steps = 1.0 / 120.0, 30
    for _ in range(steps):
        sim.step(np.zeros((1, 28)), n_sub=4, h=h)
    k = steps * 4
    g = 9.81
    # semi-implicit Euler: v_k = v0 - g k h ; z_k = z0 + sum_{i=1..k} v_i h
    vz = 1.0 - g * k * h
    z = 5.0 + h * (k * 1.0 - g * h * k * (k + 1) / 2)
    P, _, _ = momentum(model, sim)
    _, sm = model
    vcom = P / sm.total_mass
    assert abs(vcom[2] - vz) < 2e-3
    assert abs(vcom[0] - 0.3) < 2e-3 and abs(vcom[1] + 0.2) < 2e-3
    assert abs(sim.root_state[0, 2] - z) < 0.02     # root ~ com for the held pose


def _momentum_drift(model, h, n_sub, steps=60):
    rng = np.random.default_rng(0)
    sim = make(model, gravity=0.0, angular_damping=0.0)        # (link damping is an external couple: off for a conservation law)
    sim.root_state[0, 0:3] = [0.1, -0.3, 2.0]
    sim.root_state[0, 7:13] = rng.standard_normal(6) * 0.5
    sim.dof_state[0, :, 0] = rng.standard_normal(28) * 0.3
    sim.dof_state[0, :, 1] = rng.standard_normal(28) * 2.0
    sim.refresh_bodies()
    P0, L0, _ = momentum(model, sim)
    act = rng.standard_normal((1, 28)) * 0.5
    for _ in range(steps):
        sim.step(act, n_sub=n_sub, h=h)
    P1, L1, _ = momentum(model, sim)
    assert np.all(np.isfinite(sim.root_state))
    return np.abs(P1 - P0).max(), np.abs(L1 - L0).max(), np.linalg.norm(P0), np.linalg.norm(L0)


def test_momentum_conservation_zero_gravity(model):
    """Joint drives and limits are internal forces: total linear and angular momentum are constants of the
    motion.  The semi-implicit Euler step conserves them to first order: the drift over 0.5 s of violent motion
    (joint rates ~2 rad/s) is small at the production step (1/120 s) and shrinks 4x when the step does."""
    dP1, dL1, nP, nL = _momentum_drift(model, 1.0 / 120.0, 4)
    dP4, dL4, _, _ = _momentum_drift(model, 1.0 / 480.0, 16)
    assert dP1 / nP < 0.04 and dL1 / nL < 0.15
    assert 3.0 < dP1 / dP4 < 5.0 and 3.0 < dL1 / dL4 < 5.0


def test_energy_without_drives(model):
    """No gravity, no PD, no limits, no contact: kinetic energy is conserved up to the integrator's O(h) drift."""
    rng = np.random.default_rng(1)
    km, sm = model
    # (no link-link contact either: without drives and limits the limbs swing through each other, and contact damping dissipates)
    sim = make(model, gravity=0.0, limit_kp=0.0, limit_kd=0.0, angular_damping=0.0, self_collision=False)
    for d in range(28):
        sim.m.kp[d] = 0.0
        sim.m.kd[d] = 0.0
    sim.root_state[0, 2] = 2.0
    sim.root_state[0, 10:13] = rng.standard_normal(3) * 0.5
    sim.dof_state[0, :, 1] = rng.standard_normal(28) * 1.0
    sim.refresh_bodies()
    _, _, T0 = momentum(model, sim)
    for _ in range(30):
        sim.step(np.zeros((1, 28)), n_sub=8, h=1.0 / 240.0)
    _, _, T1 = momentum(model, sim)
    assert abs(T1 - T0) / T0 < 0.05


def test_stand_on_flat_ground(model):
    """PD holds the zero pose; the feet carry the weight: sum of contact forces = m g, the root stays up.
    (Checked 0.8 s after the drop: the zero pose has its centre of mass over the heels and slowly tips over
    afterwards, as a real passive humanoid would.)"""
    km, sm = model
    sim = make(model, hf=np.zeros((20, 20), np.float32))
    sim.root_state[0, 2] = 0.92
    for _ in range(25):
        sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
    assert np.all(np.isfinite(sim.root_state))
    fz = sim.contact_forces[0, :, 2].sum()
    assert abs(fz - sm.total_mass * 9.81) / (sm.total_mass * 9.81) < 0.05
    assert 0.8 < sim.root_state[0, 2] < 0.95
    feet = sim.contact_forces[0, [11, 14], 2]
    assert np.all(feet > 50.0)                          # both feet loaded
    others = np.delete(sim.contact_forces[0, :, 2], [11, 14])
    assert np.all(np.abs(others) < 1e-3)
    assert np.linalg.norm(sim.root_state[0, 7:10]) < 0.3


def test_friction_stops_sliding(model):
    km, sm = model
    sim = make(model, hf=np.zeros((40, 40), np.float32))
    sim.root_state[0, 2] = 0.9
    sim.root_state[0, 7] = 1.0      # sliding start
    for _ in range(60):
        sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
    assert abs(sim.rigid_body_state[0, 11, 7]) < 0.2   # foot no longer slides


def test_wall_blocks_motion(model):
    """A column 1 m high next to the character stops a ballistic pelvis: side contact gives a horizontal normal."""
    km, sm = model
    hf = np.zeros((40, 40), np.float32)
    hf[25:, :] = 3.0                                      # wall at x >= -4 + 24.5*0.4 = 5.8
    sim = make(model, hf=hf, gravity=0.0)
    sim.root_state[0, 0:3] = [5.0, 0.0, 1.0]             # feet just above the ground, no gravity
    sim.root_state[0, 7] = 3.0
    for _ in range(40):
        sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
    assert sim.root_state[0, 0] < 5.9                     # did not tunnel into the wall
    assert sim.root_state[0, 7] < 0.5


def test_joint_limits_hold(model):
    km, sm = model
    sim = make(model, gravity=0.0)
    for d in range(28):
        sim.m.kp[d] = 0.0
        sim.m.kd[d] = 0.0
    sim.root_state[0, 2] = 2.0
    sim.dof_state[0, 17, 1] = 8.0           # right knee (range 0..160 deg) driven into hyper-extension
    sim.dof_state[0, 17, 0] = 0.05
    sim.dof_state[0, 24, 1] = -8.0
    for _ in range(40):
        sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
    assert sim.dof_state[0, 24, 0] > -0.25
    assert sim.dof_state[0, 17, 0] < np.deg2rad(160) + 0.25


def test_random_actions_stay_finite(model):
    rng = np.random.default_rng(5)
    hf = (rng.random((40, 40)) * 0.6).astype(np.float32)
    sim = make(model, n=16, hf=hf)
    sim.root_state[:, 0:2] = rng.random((16, 2)) * 4.0
    sim.root_state[:, 2] = 1.6
    for _ in range(120):
        sim.step(rng.standard_normal((16, 28)) * 1.5, n_sub=4, h=1.0 / 120.0)
    assert np.all(np.isfinite(sim.root_state)) and np.all(np.isfinite(sim.dof_state))
    assert np.all(np.abs(sim.root_state[:, 7:13]) < 120.0)
    assert np.all(sim.root_state[:, 2] > -0.2)          # nobody fell through the ground


def test_both_formulations_agree(model):
    """The body-per-lane kernel (level-synchronous sweeps, contacts cached per lane) and the one-env-per-lane core (serial loops)
    are two independent codings of the same equations: on a contact-rich scene they must agree to fp32 reassociation error."""
    host_only()
    from oracle.sim_host import HostSim
    km, sm = model
    rng = np.random.default_rng(11)
    hf = (rng.random((40, 40)) * 0.5).astype(np.float32)
    n = 12
    sims = [HostSim(copy.deepcopy(sm.struct), n, hf, [-4.0, -4.0], [0.4, 0.4], variant=v) for v in ("core", "bpl")]
    rs = np.zeros((n, 13), np.float32)
    rs[:, 0:2] = rng.random((n, 2)) * 4.0
    rs[:, 2] = 1.0 + 0.3 * rng.random(n)
    rs[:, 6] = 1.0
    ds = (rng.standard_normal((n, 28, 2)) * 0.2).astype(np.float32)
    for s_ in sims:
        s_.root_state[:], s_.dof_state[:] = rs, ds
    hit = 0.0
    for _ in range(12):
        act = (rng.standard_normal((n, 28)) * 0.5).astype(np.float32)
        for s_ in sims:
            s_.step(act, n_sub=4, h=1.0 / 120.0)
        hit = max(hit, float(np.abs(sims[0].contact_forces).max()))
        a, b = sims
        np.testing.assert_allclose(b.root_state[:, 0:7], a.root_state[:, 0:7], atol=2e-4)
        np.testing.assert_allclose(b.dof_state[..., 0], a.dof_state[..., 0], atol=5e-4)
        np.testing.assert_allclose(b.rigid_body_state[..., 0:3], a.rigid_body_state[..., 0:3], atol=5e-4)
        np.testing.assert_allclose(b.contact_forces, a.contact_forces, atol=1.0, rtol=1e-2)     # N; stiff in the penetration depth
        for s_ in sims:                      # restart both from the same state: the comparison is per step, not of chaotic drift
            s_.root_state[:], s_.dof_state[:] = a.root_state, a.dof_state
    assert hit > 50.0                        # the scene does have contacts


def test_no_read_of_unwritten_work_memory(model):
    """Per-env work arrays (the core's Scratch / State, the kernel's LDS exchange buffer and contact cache) filled with NaN
    instead of zero must not change a single bit of the result: nothing reads an element the algorithm has not written."""
    host_only()
    import os
    import subprocess
    from oracle import sim_host
    from oracle.sim_host import HostSim
    here = os.path.dirname(os.path.abspath(sim_host.__file__))
    subprocess.check_call(["make", "-C", here, "sanitize"], stdout=subprocess.DEVNULL)
    code = r"""
import copy, sys, numpy as np
sys.path.insert(0, %r)
from oracle import sim_host
from parc_amd.anim.kin_char_model import KinCharModel
from parc_amd.assets import humanoid_spec
from parc_amd.sim_model import SimModel
km = KinCharModel("cpu"); km.load_char_file(humanoid_spec.write_mjcf()); sm = SimModel(km)
out = {}
for variant in ("core", "bpl"):
    for fill in (0x00, 0xFF):
        sim_host.set_fill(fill)
        rng = np.random.default_rng(4)
        hf = (rng.random((40, 40)) * 0.5).astype(np.float32)
        sim = sim_host.HostSim(copy.deepcopy(sm.struct), 6, hf, [-4.0, -4.0], [0.4, 0.4], variant=variant)
        sim.root_state[:, 0:2] = rng.random((6, 2)) * 4.0
        sim.root_state[:, 2] = 1.0
        for _ in range(20):
            sim.step(rng.standard_normal((6, 28)).astype(np.float32), n_sub=4, h=1.0 / 120.0)
        out[(variant, fill)] = np.concatenate([sim.root_state.ravel(), sim.dof_state.ravel(), sim.rigid_body_state.ravel(), sim.contact_forces.ravel()])
        assert np.all(np.isfinite(out[(variant, fill)])), (variant, fill)
    assert np.array_equal(out[(variant, 0x00)], out[(variant, 0xFF)]), variant
    assert np.abs(sim.contact_forces).max() > 50.0
print("identical")
""" % os.path.dirname(here)
    env = dict(os.environ, PARC_SIM_HOST_LIB=os.path.join(here, "_build", "libparc_sim_host_poison.so"))
    res = subprocess.run([__import__("sys").executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and "identical" in res.stdout, res.stderr[-3000:]


@pytest.mark.parametrize("clip,terrain,max_pen", [(0, "g5_hf_civ", 0.002), (1, "g5_hf_teaser", 0.03)])
def test_kinematic_replay_of_the_shipped_clips_does_not_penetrate(model, clip, terrain, max_pen):
    """the authors' two clips (data/terrains/civilization.pkl, TEASER_TERRAIN.pkl; frames in fixture G3, heightfields in
    G5) replayed kinematically on their own terrains.  The collision geometry built from the MJCF (sample spheres on every geom, box
    corners for the feet) against the column terrain must report essentially no penetration, contacts on the feet only, and none
    at all in frames the clip labels as airborne -- evidence that geometry and contact detection agree with the data the reference
    was built around, independent of Isaac Gym."""
    host_only()
    from conftest import golden
    from oracle.sim_host import HostSim
    km, sm = model
    z3, z5 = golden("g3_motion"), golden(terrain)
    fr, con = z3["frames_%d" % clip], z3["contacts_%d" % clip]
    F = fr.shape[0]
    sim = HostSim(copy.deepcopy(sm.struct), F, z5["hf"], z5["min_point"], z5["dxdy"])
    ang = np.linalg.norm(fr[:, 3:6], axis=-1, keepdims=True)
    sim.root_state[:, 0:3] = fr[:, 0:3]
    sim.root_state[:, 3:6] = fr[:, 3:6] * np.where(ang > 1e-8, np.sin(ang / 2) / np.maximum(ang, 1e-8), 0.5)
    sim.root_state[:, 6] = np.cos(ang[:, 0] / 2)
    sim.dof_state[:, :, 0] = fr[:, 6:]
    pen = sim.penetration()
    body = np.array(list(sm.struct.sph_body)[:sm.struct.num_spheres])
    # measured: 0.6 mm over the 254 frames of the civilization clip; the 58-frame teaser clip (a kinematic generator's output, a drop
    # onto a lower platform) sinks its feet by up to 2.4 cm in the 3-4 frames after each of its three landings, 0 elsewhere
    assert pen.max() < max_pen, pen.max()
    assert np.mean(pen.max(axis=1) > 0.01) < 0.2
    # (a touch = more than 1 mm: the clips' feet graze the surface within fp32 noise in a few more frames)
    touching = np.stack([(pen[:, body == b].max(axis=1) > 1e-3) if np.any(body == b) else np.zeros(F, bool) for b in range(15)], axis=1)
    assert not (pen[:, ~np.isin(body, [11, 14])] > 0).any()
    feet = [km.get_body_id("right_foot"), km.get_body_id("left_foot")]
    assert not touching[:, [b for b in range(15) if b not in feet]].any()       # only the feet ever reach the ground
    labelled = con > 0.5
    near = labelled.copy()                                   # the clip's own contact labels, widened by two frames (they are per-frame
    for k in (1, 2):                                         # annotations of a 30 fps clip; the geometry can touch a frame earlier / later)
        near[k:] |= labelled[:-k]
        near[:-k] |= labelled[k:]
    airborne = ~near.any(axis=1)
    assert airborne.sum() >= 0 and not touching[airborne].any()      # no contact (hence no contact force) where the clip is airborne
    for b in feet:                                           # a geometric touch happens only where the clip says that foot is in contact
        assert not (touching[:, b] & ~near[:, b]).any(), np.nonzero(touching[:, b] & ~near[:, b])
    assert (pen > 0).any()


def spin_momentum(model, sim, e=0):
    """sum over the links of I_com,i w_i (world frame): the part of the angular momentum that link damping acts on"""
    _, sm = model
    S = np.zeros(3)
    for b in range(sim.B):
        bs = sim.rigid_body_state[e, b].astype(np.float64)
        R = rotm(bs[3:7])
        S += R @ sm.body_inertia_com[b] @ R.T @ bs[10:13]
    return S


def test_link_angular_damping_is_a_couple_on_every_link(model):
    """asset_options.angular_damping = 0.01 (envs/ig_char_env.py:141): every link feels the couple -c I_com,i w_i (a free body then
    obeys dw/dt = -c w).  For the whole character, held rigid by its drives and spinning freely (no gravity, no contact), the
    total angular momentum therefore changes at the rate dL/dt = -c sum_i I_com,i w_i -- not at -c L: the orbital part of L is
    untouched -- and not at all without damping."""
    km, sm = model
    c, T = 0.05, 1.0
    res = {}
    for damp in (0.0, c):
        sim = make(model, gravity=0.0, angular_damping=damp)
        sim.root_state[0, 2] = 3.0
        sim.root_state[0, 10:13] = [0.3, -0.2, 1.5]
        sim.refresh_bodies()
        _, L0, _ = momentum(model, sim)
        S0 = spin_momentum(model, sim)
        for _ in range(int(T * 30)):
            sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
        _, L1, _ = momentum(model, sim)
        res[damp] = (L0, L1, S0, spin_momentum(model, sim))
    L0, L1_free, S0, S1 = res[0.0]
    assert np.linalg.norm(L1_free - L0) / np.linalg.norm(L0) < 1e-2          # the integrator's own O(h) drift, no damping
    dL = res[c][1] - L1_free                                                  # what the damping did
    expect = -c * T * 0.5 * (S0 + S1)
    assert np.linalg.norm(dL - expect) < 0.1 * np.linalg.norm(expect), (dL, expect)
    assert np.linalg.norm(S0) < 0.5 * np.linalg.norm(L0)                      # (most of L is orbital: the decay is slower than e^{-ct})
    assert sm.struct.angular_damping == pytest.approx(0.01) and sm.struct.max_angular_velocity == pytest.approx(100.0)


# ---------------------------------------------------------------------------------------------------------------
# link-link contact (self-collision): Isaac Gym creates the character with collision filter 0 (envs/ig_char_env.py:105-113),
# so links that are not joined by a joint collide
# ---------------------------------------------------------------------------------------------------------------
def capsule_gap(model, sim, bi, bj, e=0):
    """distance between the capsule surfaces of two bodies (negative = overlapping)"""
    _, sm = model

    def world(b):
        bs = sim.rigid_body_state[e, b].astype(np.float64)
        R = rotm(bs[3:7])
        p0, p1, r = sm.capsules[b]
        return bs[0:3] + R @ p0, bs[0:3] + R @ p1, r
    a0, a1, ra = world(bi)
    b0, b1, rb = world(bj)
    ts = np.linspace(0, 1, 41)
    pa = a0[None] + ts[:, None] * (a1 - a0)[None]
    pb = b0[None] + ts[:, None] * (b1 - b0)[None]
    return np.linalg.norm(pa[:, None] - pb[None], axis=-1).min() - ra - rb


def test_self_collision_set_and_rest_poses(model):
    """Every pair of bodies not joined by a joint is in the collision set (91 pairs for the 15-body humanoid), symmetric; the zero
    pose and the authors' two clips are free of link-link contact (no contact force when replayed away from any terrain)."""
    from conftest import golden
    from oracle.sim_host import HostSim
    km, sm = model
    par = [int(p) for p in km._parent_indices]
    pairs = 0
    for b in range(15):
        for j in range(15):
            on = bool((sm.struct.self_mask[b] >> j) & 1)
            assert on == bool((sm.struct.self_mask[j] >> b) & 1)
            assert on == (j != b and par[b] != j and par[j] != b)
            pairs += on
    assert pairs == 2 * 91 and all(r > 0 for _, _, r in sm.capsules)
    z3 = golden("g3_motion")
    for clip in (0, 1):
        fr = z3["frames_%d" % clip]
        F = fr.shape[0]
        sim = HostSim(copy.deepcopy(sm.struct), F, np.full((20, 20), -100.0, np.float32), [-4.0, -4.0], [0.4, 0.4])
        ang = np.linalg.norm(fr[:, 3:6], axis=-1, keepdims=True)
        sim.root_state[:, 0:3] = fr[:, 0:3]
        sim.root_state[:, 3:6] = fr[:, 3:6] * np.where(ang > 1e-8, np.sin(ang / 2) / np.maximum(ang, 1e-8), 0.5)
        sim.root_state[:, 6] = np.cos(ang[:, 0] / 2)
        sim.dof_state[:, :, 0] = fr[:, 6:]
        sim.step(fr[:, 6:].copy(), n_sub=1, h=1.0 / 120.0)
        assert np.abs(sim.contact_forces).max() == 0.0
    sim = make(model)
    sim.root_state[0, 2] = 5.0
    sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
    assert np.abs(sim.contact_forces).max() == 0.0


def test_self_collision_keeps_crossing_legs_apart(model):
    """Both hips are driven into adduction so that the legs cross.  Without link-link contact the thigh capsules end up 11 cm inside
    each other; with it they are held at the surface (about a centimetre of spring compression against the drives), both bodies report the
    contact force, and the two forces are equal and opposite."""
    km, sm = model
    rt, lt, rs, ls = [km.get_body_id(n) for n in ("right_thigh", "left_thigh", "right_shin", "left_shin")]
    res = {}
    for on in (True, False):
        sim = make(model, gravity=0.0, self_collision=on)
        sim.root_state[0, 2] = 3.0
        act = np.zeros((1, 28), np.float32)
        act[0, 14] = 0.9                     # right hip, rotation about x
        act[0, 21] = -0.9                    # left hip, mirrored
        gaps, forces = [], []
        for _ in range(45):
            sim.step(act, n_sub=4, h=1.0 / 120.0)
            gaps.append(min(capsule_gap(model, sim, i, j) for i, j in ((rt, lt), (rs, ls), (rt, ls), (rs, lt))))
            forces.append(sim.contact_forces[0].copy())
        res[on] = (np.array(gaps), np.array(forces))
    gap_on, f_on = res[True]
    gap_off, f_off = res[False]
    assert gap_off.min() < -0.08                      # without contact: the legs pass through each other
    assert gap_on.min() > -0.03 and gap_on[-1] > -0.015       # with contact: a brief overshoot at impact, then held near the surface
    # (the pair's explicit spring is limited by the links' reduced mass, here 34 kN/m: ~1 cm under the ~300 N of the two hip drives)
    assert np.abs(f_off).max() == 0.0
    legs = [rt, lt, rs, ls, km.get_body_id("right_foot"), km.get_body_id("left_foot")]
    assert np.linalg.norm(f_on[:, legs], axis=-1).max() > 50.0
    # action = reaction: the contact forces of the whole character sum to zero in every frame (there is no terrain in this scene)
    tot = f_on.sum(axis=1)
    assert np.abs(tot).max() < 2e-3 * max(np.abs(f_on).max(), 1.0)


def test_momentum_is_conserved_through_link_contacts(model):
    """Link-link contact forces are internal: with the arms driven into the body in zero gravity, linear and angular momentum keep
    their values to the integrator's O(h) accuracy, exactly as without contacts."""
    km, sm = model
    sim = make(model, gravity=0.0, angular_damping=0.0)
    rng = np.random.default_rng(2)
    sim.root_state[0, 0:3] = [0.0, 0.0, 3.0]
    sim.root_state[0, 7:13] = rng.standard_normal(6) * 0.3
    sim.refresh_bodies()
    P0, L0, _ = momentum(model, sim)
    act = np.zeros((1, 28), np.float32)
    act[0, 14] = 0.9                         # the legs are driven across each other
    act[0, 21] = -0.9
    touched = False
    for _ in range(60):
        sim.step(act, n_sub=4, h=1.0 / 120.0)
        touched |= bool(np.abs(sim.contact_forces).max() > 1.0)
    P1, L1, _ = momentum(model, sim)
    assert touched
    # (2.0 % with and 2.2 % without the link contacts at h = 1/120, 0.5 % / 0.6 % at h = 1/480: the O(h) drift of a semi-implicit Euler step in
    # reduced coordinates - P = J(q) qdot is advanced at the old configuration -, the same with and without contacts)
    assert np.abs(P1 - P0).max() < 0.03 * max(np.linalg.norm(P0), 1.0)
    assert np.abs(L1 - L0).max() < 0.05 * max(np.linalg.norm(L0), 1.0)


def test_link_contact_forces_are_equal_and_opposite_for_parallel_capsules(model):
    """The degenerate case of the closest-point computation: EXACTLY parallel, overlapping capsules (thighs and shins in the rest pose,
    radii enlarged until they overlap).  seg_seg_closest clamps s = 0 on its FIRST segment there, so the two bodies of a pair must
    evaluate it in the same (canonical) order: the reported contact forces of the two bodies then cancel to rounding, in every substep,
    and momentum is conserved like without contact.  (With each body passing its own capsule first the two lanes picked different
    closest-point pairs - round-2 advisor finding.)"""
    km, sm = model
    sim = make(model, gravity=0.0, angular_damping=0.0)
    for b in (9, 10, 12, 13):                  # right / left thigh and shin: axes parallel, 0.17 m apart
        sim.m.cap_radius[b] = 0.095
    rng = np.random.default_rng(5)
    sim.root_state[0, 0:3] = [0.0, 0.0, 3.0]
    sim.root_state[0, 7:13] = rng.standard_normal(6) * 0.2
    sim.refresh_bodies()
    P0, L0, _ = momentum(model, sim)
    act = np.zeros((1, 28), np.float32)
    worst, peak = 0.0, 0.0
    for _ in range(40):
        sim.step(act, n_sub=1, h=1.0 / 120.0)   # one substep per call: contact_forces is that substep's force, not a mean
        f = sim.contact_forces[0].astype(np.float64)
        peak = max(peak, float(np.abs(f).max()))
        worst = max(worst, float(np.abs(f.sum(axis=0)).max()))
        # pairwise: thigh against thigh, shin against shin (nothing else touches in this pose)
        assert np.abs(f[9] + f[12]).max() <= 1e-4 * max(np.abs(f[9]).max(), 1.0)
        assert np.abs(f[10] + f[13]).max() <= 1e-4 * max(np.abs(f[10]).max(), 1.0)
    assert peak > 20.0                          # they do press on each other
    assert worst <= 1e-4 * peak                 # internal forces sum to zero
    P1, L1, _ = momentum(model, sim)
    assert np.abs(P1 - P0).max() < 0.02 * max(np.linalg.norm(P0), 1.0)
    assert np.abs(L1 - L0).max() < 0.05 * max(np.linalg.norm(L0), 1.0)


def test_crossed_legs_do_not_slide_freely(model):
    """Friction between links (Isaac Gym: collision filter 0, envs/ig_char_env.py:105-113 - link-link contacts carry the material's
    friction): the thighs are pressed against each other by the hip drives, then driven along each other.  With mu = 1 the sliding
    is held back while they touch - the hips travel clearly less in the first steps than with mu = 0 - momentum stays conserved (the
    friction forces are an equal and opposite pair on one line of action), and with mu = 0 the contact is the frictionless one."""
    km, sm = model

    def run(mu):
        sim = make(model, gravity=0.0, angular_damping=0.0, friction_mu=mu)
        sim.root_state[0, 0:3] = [0.0, 0.0, 3.0]
        sim.refresh_bodies()
        P0, L0, _ = momentum(model, sim)
        act = np.zeros((1, 28), np.float32)
        act[0, 14], act[0, 21] = 0.3, -0.3                  # hips adducted: the legs press on each other
        for _ in range(45):
            sim.step(act, n_sub=4, h=1.0 / 120.0)
        pressed = float(np.linalg.norm(sim.contact_forces[0], axis=-1).max())
        act[0, 15], act[0, 22] = -0.3, 0.3                  # ... and are now driven along each other (hip flexion, opposite signs)
        travel = []
        for _ in range(4):
            sim.step(act, n_sub=4, h=1.0 / 120.0)
            travel.append(0.5 * (abs(float(sim.dof_state[0, 15, 0])) + abs(float(sim.dof_state[0, 22, 0]))))
        P1, L1, _ = momentum(model, sim)
        return pressed, travel, max(np.abs(P1 - P0).max(), np.abs(L1 - L0).max())
    p0, t0, dm0 = run(0.0)
    p1, t1, dm1 = run(1.0)
    assert p0 > 50.0 and p1 > 50.0                          # pressed together before the slide starts
    assert t1[1] < 0.7 * t0[1] and t1[3] < 0.8 * t0[3]      # held back (0.028 / 0.052 rad after two steps, 0.12 / 0.18 after four)
    assert t1[3] > 0.02                                     # regularised Coulomb friction: it slides, it is not glued
    print("momentum drift without / with friction:", dm0, dm1)
    assert dm1 < 1.5 * dm0 + 0.05                           # internal forces either way: no drift beyond the integrator's own (mu = 0 run)
    # (mu is the only difference between the two runs, and the frictionless one is what round 2 shipped)
    assert np.abs(np.array(t0) - np.array(t1)).max() > 0.02


def test_forward_dynamics_against_independent_inverse_dynamics(model):
    """The simulator's forward dynamics (Featherstone's articulated-body algorithm in reduced coordinates, fp32) checked against an
    INDEPENDENT formulation: recursive Newton-Euler inverse dynamics written here in numpy float64 from the textbook vector
    equations (world-frame velocities / accelerations body by body, then forces leaf to root).  One tiny step (h = 1e-4 s) from a
    random state with the drives, limits, damping and contacts off gives the accelerations by finite differences; fed to the
    inverse dynamics they must need no joint torque beyond what the armature (rotor inertia) takes, and no wrench on the free root.
    Isaac Gym's arithmetic is unavailable, so this is the strongest statement available that the equations of motion are the right
    ones: frames, joint conventions, inertia composition, velocity-product terms, gravity."""
    km, sm = model
    rng = np.random.default_rng(7)
    h = 1e-4
    sim = make(model, angular_damping=0.0, limit_kp=0.0, limit_kd=0.0, self_collision=False)
    for d in range(28):
        sim.m.kp[d] = 0.0
        sim.m.kd[d] = 0.0
    B = 15
    q = rng.standard_normal(4)
    sim.root_state[0, 0:3] = [0.3, -0.2, 5.0]
    sim.root_state[0, 3:7] = q / np.linalg.norm(q)
    sim.root_state[0, 7:13] = rng.standard_normal(6) * 1.0
    sim.dof_state[0, :, 0] = rng.standard_normal(28) * 0.4
    sim.dof_state[0, :, 1] = rng.standard_normal(28) * 2.0
    sim.refresh_bodies()
    rs0, ds0, rb0 = sim.root_state[0].astype(np.float64), sim.dof_state[0].astype(np.float64), sim.rigid_body_state[0].astype(np.float64)
    sim.step(np.zeros((1, 28)), n_sub=1, h=h)
    rs1, ds1 = sim.root_state[0].astype(np.float64), sim.dof_state[0].astype(np.float64)
    # ---- inverse dynamics at the initial state with the finite-difference accelerations
    s = sm.struct
    par = [int(s.parent[b]) for b in range(B)]
    R = [rotm(rb0[b, 3:7]) for b in range(B)]
    P = [rb0[b, 0:3] for b in range(B)]
    g = np.array([0.0, 0.0, -float(s.gravity)])
    w, al, a = [None] * B, [None] * B, [None] * B
    w[0] = rs0[10:13]
    al[0] = (rs1[10:13] - rs0[10:13]) / h
    a[0] = (rs1[7:10] - rs0[7:10]) / h                     # acceleration of the root origin, world frame
    wj_all, wjd_all = {}, {}
    for b in range(1, B):
        jt, d0 = int(s.joint_type[b]), int(s.dof_idx[b])
        if jt == 2:                                        # spherical: child-frame angular velocity components
            wj, wjd = ds0[d0:d0 + 3, 1], (ds1[d0:d0 + 3, 1] - ds0[d0:d0 + 3, 1]) / h
        elif jt == 1:                                      # hinge: rate about the joint axis
            ax = np.array([s.joint_axis[b][k] for k in range(3)])
            wj, wjd = ds0[d0, 1] * ax, (ds1[d0, 1] - ds0[d0, 1]) / h * ax
        else:
            wj, wjd = np.zeros(3), np.zeros(3)
        wj_all[b], wjd_all[b] = wj, wjd
        p = par[b]
        r = P[b] - P[p]
        w[b] = w[p] + R[b] @ wj
        al[b] = al[p] + R[b] @ wjd + np.cross(w[p], R[b] @ wj)
        a[b] = a[p] + np.cross(al[p], r) + np.cross(w[p], np.cross(w[p], r))
    f, n = [None] * B, [None] * B
    for b in range(B):
        c = R[b] @ sm.body_com[b]
        Iw = R[b] @ sm.body_inertia_com[b] @ R[b].T
        ac = a[b] + np.cross(al[b], c) + np.cross(w[b], np.cross(w[b], c))
        F = sm.body_mass[b] * (ac - g)
        N = Iw @ al[b] + np.cross(w[b], Iw @ w[b])
        f[b], n[b] = F, N + np.cross(c, F)                 # moment about the body origin
    for b in range(B - 1, 0, -1):
        p = par[b]
        f[p] = f[p] + f[b]
        n[p] = n[p] + n[b] + np.cross(P[b] - P[p], f[b])
    scale = sm.total_mass * 9.81 * 0.5                     # ~ weight x half a metre: the natural torque scale (245 N m)
    # measured: 0.10 N of 490 N on the root, 0.015 N m root moment, 0.008 N m worst joint torque (fp32 finite differences at h = 1e-4)
    assert np.linalg.norm(f[0]) < 5e-4 * sm.total_mass * 9.81, f[0]        # no net force on a free-floating character
    assert np.linalg.norm(n[0]) < 5e-4 * scale, n[0]
    worst = 0.0
    for b in range(1, B):
        jt, d0 = int(s.joint_type[b]), int(s.dof_idx[b])
        tau = R[b].T @ n[b]                                # joint torque in the child frame
        if jt == 2:
            arm = np.array([s.armature[d0 + k] for k in range(3)])
            res = tau + arm * (ds1[d0:d0 + 3, 1] - ds0[d0:d0 + 3, 1]) / h
        elif jt == 1:
            ax = np.array([s.joint_axis[b][k] for k in range(3)])
            res = np.array([ax @ tau + s.armature[d0] * (ds1[d0, 1] - ds0[d0, 1]) / h])
        else:
            continue                                       # a fixed joint transmits any torque
        worst = max(worst, float(np.abs(res).max()))
    print('RNEA residuals: root force %.3e N, root moment %.3e N m, worst joint torque %.3e N m (scale %.0f N m)' % (np.linalg.norm(f[0]), np.linalg.norm(n[0]), worst, scale))
    assert worst < 5e-4 * scale, worst
    # the check has teeth: the same residual with the velocity-product terms left out is two orders of magnitude larger
    assert max(float(np.linalg.norm(np.cross(w[b], (R[b] @ sm.body_inertia_com[b] @ R[b].T) @ w[b]))) for b in range(B)) > 0.05


# ---------------------------------------------------------------------------------------------------------------------------------
# Contact closed forms on the smallest model the kernels accept: ONE free body, a uniform solid sphere with one collision sphere
# (reference configuration of the material: friction 1 / 1, restitution 0 - util/ig_util.py:6-22, envs/ig_env.py:131-164).
# ---------------------------------------------------------------------------------------------------------------------------------
def make_ball(model, mass=10.0, radius=0.2, hf=None, **over):
    km, sm = model
    s = copy.deepcopy(sm.struct)
    s.num_bodies, s.dof_size, s.num_spheres = 1, 0, 1
    s.parent[0], s.joint_type[0], s.dof_idx[0] = -1, int(sm.struct.joint_type[0]), 0
    s.mass[0] = mass
    for k in range(3):
        s.com[0][k] = 0.0
        s.sph_pos[0][k] = 0.0
    i_s = 0.4 * mass * radius * radius
    for k, v in enumerate([i_s, 0.0, 0.0, i_s, 0.0, i_s]):          # (xx xy xz yy yz zz) about the body origin = the centre
        s.inertia_o[0][k] = v
    s.sph_body[0], s.sph_radius[0] = 0, radius
    for b in range(16):
        s.self_mask[b] = 0
        s.cap_radius[b] = 0.0
    s.angular_damping = 0.0
    for k, v in over.items():
        setattr(s, k, v)
    if hf is None:
        hf = np.zeros((60, 20), np.float32)
    if _FORMULATION[0] == "device":
        from device_sim import DeviceSim
        return DeviceSim(s, 1, hf, [-4.0, -4.0], [0.4, 0.4], num_bodies=1, dof_size=0)
    from oracle.sim_host import HostSim
    return HostSim(s, 1, hf, [-4.0, -4.0], [0.4, 0.4], num_bodies=1, dof_size=0)


def test_ball_rests_at_the_penalty_springs_closed_form_depth(model):
    """A body at rest on a flat column top: the contact spring carries the weight, k d = m g, so the centre sits at
    z = r - m g / k (98.1 N / 4e4 N/m = 2.45 mm), the reported net contact force is (0, 0, m g) and nothing moves."""
    mass, r = 10.0, 0.2
    sim = make_ball(model, mass, r)
    kn, g = float(sim.m.contact_kn), float(sim.m.gravity)
    sim.root_state[0, 0:3] = [0.0, 0.0, r + 0.01]
    act = np.zeros((1, 0), np.float32)
    for _ in range(120):
        sim.step(act, n_sub=4, h=1.0 / 120.0)
    d = mass * g / kn
    assert d < float(sim.m.contact_max_pen)
    assert abs(sim.root_state[0, 2] - (r - d)) < 2e-5, (sim.root_state[0, 2], r - d)
    assert np.abs(sim.root_state[0, 7:13]).max() < 1e-4
    f = sim.contact_forces[0, 0]
    assert abs(f[2] - mass * g) < 1e-3 * mass * g and np.abs(f[0:2]).max() < 1e-3


def test_sliding_ball_decelerates_at_mu_g_then_rolls_at_five_sevenths(model):
    """Coulomb friction, mu = 1: a ball set sliding without spin at v0 decelerates at mu g while it slips (v = v0 - mu g t), spins up
    under the friction couple (r w = 5/2 mu g t for a uniform sphere) and ends up ROLLING at 5/7 v0 - whatever the friction law does
    in between, because the friction force acts at the contact point and the angular momentum about it is conserved
    (m v0 r = m v r + 2/5 m r^2 v / r).  Slip ends at t* = 2 v0 / (7 mu g)."""
    mass, r, v0 = 10.0, 0.2, 3.0
    sim = make_ball(model, mass, r)
    mu, g, kn = float(sim.m.friction_mu), float(sim.m.gravity), float(sim.m.contact_kn)
    assert mu == 1.0
    sim.root_state[0, 0:3] = [-3.0, 0.0, r - mass * g / kn]          # at its rest depth: no bounce
    sim.root_state[0, 7] = v0
    act = np.zeros((1, 0), np.float32)
    h = 1.0 / 120.0
    t_star = 2 * v0 / (7 * mu * g)                                    # 0.087 s = 10.5 substeps... use finer substeps to resolve it
    hs = h / 4
    k1 = int(0.5 * t_star / hs)
    for _ in range(k1):
        sim.step(act, n_sub=1, h=hs)
    t1 = k1 * hs
    # (the friction is implicit in the slip: F = -mu N v_slip(end of substep) / v_slip(start), up to 5 % inside the cone at these speeds)
    assert abs(sim.root_state[0, 7] - (v0 - mu * g * t1)) < 0.01 * v0, (sim.root_state[0, 7], v0 - mu * g * t1)
    assert abs(r * sim.root_state[0, 11] - 2.5 * mu * g * t1) < 0.02 * v0, (r * sim.root_state[0, 11], 2.5 * mu * g * t1)
    f = sim.contact_forces[0, 0]
    assert -1.0005 * mu * mass * g < f[0] < -0.93 * mu * mass * g           # on (never outside) the friction cone, opposing the slip
    assert abs(f[2] - mass * g) < 1e-3 * mass * g                          # the spin does not disturb the normal force
    assert abs(sim.root_state[0, 2] - (r - mass * g / kn)) < 2e-5 and abs(sim.root_state[0, 9]) < 1e-4
    for _ in range(240):
        sim.step(act, n_sub=1, h=hs)
    v, w = sim.root_state[0, 7], sim.root_state[0, 11]
    assert abs(v - 5.0 / 7.0 * v0) < 0.003 * v0, (v, 5.0 / 7.0 * v0)
    assert abs(r * w - v) < 0.003 * v0                                  # rolling without slipping (about +y for motion along +x)
    assert abs(sim.contact_forces[0, 0, 0]) < 0.02 * mass * g          # no friction force left once it rolls
    assert np.abs(sim.root_state[0, [8, 9, 10, 12]]).max() < 1e-3


def test_ball_thrown_at_a_wall_loses_its_normal_speed_and_leaves_rolling(model):
    """Restitution 0 (envs/ig_env.py:517,733, tracker_config/dm_env_default.yaml:126) and Coulomb friction at a side contact (a column
    wall: horizontal normal).  A ball thrown at a wall at 2 m/s with 0.5 m/s of upward slip, no gravity:
    * the spring-damper contact works on approach AND on rebound (the force is only clamped at zero: no adhesion), so the ball comes
      back with a fraction of its approach speed - 0.32 here: one contact point under a 10 kg ball is under-damped, zeta = 0.79, and a
      1/120 s substep resolves the 63 rad/s contact coarsely; with the damper on approach only (rounds 1-3) it came back at 0.73 -, and
      a ball DROPPED on the floor from 0.2 m does not leave it again at all;
    * the friction impulse it needs to start rolling up the wall, (2/7) m 0.5, is far inside the cone mu x (normal impulse), so it leaves
      ROLLING: v_z = 5/7 x 0.5 and r w_y = v_z, exactly as on the floor."""
    mass, r = 10.0, 0.2
    hf = np.zeros((60, 20), np.float32)
    hf[30:, :] = 5.0                                                   # wall face at x = -4 + 29.5 * 0.4 = 7.8
    sim = make_ball(model, mass, r, hf=hf, gravity=0.0)
    sim.root_state[0, 0:3] = [7.8 - r - 0.05, 0.0, 2.0]
    sim.root_state[0, 7:10] = [2.0, 0.0, 0.5]
    act = np.zeros((1, 0), np.float32)
    deepest = 0.0
    for _ in range(60):
        sim.step(act, n_sub=1, h=1.0 / 120.0)
        deepest = max(deepest, float(sim.root_state[0, 0]) - (7.8 - r))
    assert 0.0 < deepest < float(sim.m.contact_max_pen)               # touched, never deeper than the penetration cap
    assert -0.4 * 2.0 < sim.root_state[0, 7] < 0.0                    # comes back with less than 0.4 of the approach speed
    assert abs(sim.root_state[0, 9] - 5.0 / 7.0 * 0.5) < 2e-3          # rolling up the wall
    assert abs(r * sim.root_state[0, 11] - sim.root_state[0, 9]) < 2e-3
    sim = make_ball(model, mass, r)
    sim.root_state[0, 0:3] = [0.0, 0.0, r + 0.2]
    zs = []
    for _ in range(240):
        sim.step(act, n_sub=1, h=1.0 / 120.0)
        zs.append(float(sim.root_state[0, 2]))
    first = int(np.argmax(np.array(zs) < r))                          # first touch
    assert max(zs[first:]) < r                                         # dropped from 0.2 m: the centre never rises above r again
    assert abs(zs[-1] - (r - mass * float(sim.m.gravity) / float(sim.m.contact_kn))) < 2e-5


def test_tumbling_free_body_keeps_its_velocity(model):
    """No force, no gravity: a body that spins at 6 rad/s and drifts at 3 m/s keeps both, exactly, and travels in a straight line.
    (Advancing the BODY coordinates of the root's linear velocity by the spatial acceleration turns them by (1 - h [w]) per substep and
    stretched the speed by 16 % in one second; the root's linear velocity is advanced in world coordinates since round 4.)"""
    sim = make_ball(model, 10.0, 0.2, gravity=0.0, hf=np.full((60, 20), -100.0, np.float32))
    sim.root_state[0, 0:3] = [-3.0, 0.0, 1.0]
    sim.root_state[0, 7:10] = [3.0, 0.5, -0.2]
    sim.root_state[0, 10:13] = [1.0, 6.0, -2.0]
    act = np.zeros((1, 0), np.float32)
    for _ in range(30):
        sim.step(act, n_sub=4, h=1.0 / 120.0)
    np.testing.assert_allclose(sim.root_state[0, 7:10], [3.0, 0.5, -0.2], atol=2e-4)
    np.testing.assert_allclose(sim.root_state[0, 10:13], [1.0, 6.0, -2.0], atol=2e-4)      # a sphere's inertia is isotropic: no precession
    np.testing.assert_allclose(sim.root_state[0, 0:3], [0.0, 0.5, 0.8], atol=5e-4)
    # the same for the humanoid's centre of mass with a tumbling root (momentum() reads the published body states)
    sim = make(model, gravity=0.0, angular_damping=0.0, self_collision=False)
    sim.root_state[0, 0:3] = [0.0, 0.0, 5.0]
    sim.root_state[0, 7:10] = [2.0, 0.0, 0.0]
    sim.root_state[0, 10:13] = [0.0, 5.0, 1.0]
    sim.refresh_bodies()
    P0, _, _ = momentum(model, sim)
    for _ in range(30):
        sim.step(np.zeros((1, 28)), n_sub=4, h=1.0 / 120.0)
    P1, _, _ = momentum(model, sim)
    # (0.5 % in one second: the drives hold the pose against the centrifugal load, the configuration moves a little, and a reduced-
    # coordinate semi-implicit step conserves P = J(q) qdot to O(h) only - see test_momentum_is_conserved_through_link_contacts)
    assert np.abs(P1 - P0).max() < 1e-2 * np.linalg.norm(P0), (P0, P1)


def make_arm(model, joint, kp=40.0, kd=3.0, armature=0.02, effort=0.0, **over):
    """A two-body chain in free space without gravity: a 10^6 kg base (the free root) and one link on a hinge about y (joint=1) or on
    a spherical joint (joint=2) at the base's origin; the link: 2 kg, centre of mass 0.3 m below the joint, inertia about its centre
    diag(0.05, 0.05, 0.01) -> 0.23 kg m^2 about the joint's x and y axes.  The base is heavy enough that its reaction moves the joint
    by a millionth of the link's motion."""
    km, sm = model
    s = copy.deepcopy(sm.struct)
    D = 1 if joint == 1 else 3
    s.num_bodies, s.dof_size, s.num_spheres = 2, D, 1
    s.parent[0], s.dof_idx[0] = -1, 0
    s.parent[1], s.joint_type[1], s.dof_idx[1] = 0, joint, 0
    for k in range(3):
        s.com[0][k] = 0.0
        s.sph_pos[0][k] = 0.0
        s.local_translation[1][k] = 0.0
        s.joint_axis[1][k] = [0.0, 1.0, 0.0][k]
        s.com[1][k] = [0.0, 0.0, -0.3][k]
    for k in range(4):
        s.local_rotation[1][k] = [0.0, 0.0, 0.0, 1.0][k]
    s.mass[0], s.mass[1] = 1.0e6, 2.0
    for k, v in enumerate([1.0e6, 0.0, 0.0, 1.0e6, 0.0, 1.0e6]):
        s.inertia_o[0][k] = v
    for k, v in enumerate([0.05 + 2.0 * 0.09, 0.0, 0.0, 0.05 + 2.0 * 0.09, 0.0, 0.01]):       # about the joint (parallel axes)
        s.inertia_o[1][k] = v
    for d in range(D):
        s.kp[d], s.kd[d], s.armature[d], s.effort[d] = kp, kd, armature, effort
        s.limit_lo[d], s.limit_hi[d] = -10.0, 10.0
    s.sph_body[0], s.sph_radius[0] = 0, 0.01
    for b in range(16):
        s.self_mask[b] = 0
        s.cap_radius[b] = 0.0
    s.angular_damping, s.gravity = 0.0, 0.0
    for k, v in over.items():
        setattr(s, k, v)
    hf = np.full((20, 20), -100.0, np.float32)
    if _FORMULATION[0] == "device":
        from device_sim import DeviceSim
        sim = DeviceSim(s, 1, hf, [-4.0, -4.0], [0.4, 0.4], num_bodies=2, dof_size=D)
    else:
        from oracle.sim_host import HostSim
        sim = HostSim(s, 1, hf, [-4.0, -4.0], [0.4, 0.4], num_bodies=2, dof_size=D)
    sim.root_state[0, 0:3] = [0.0, 0.0, 5.0]
    return sim


@pytest.mark.parametrize("joint", [1, 2])
def test_joint_drive_is_the_implicit_euler_spring_damper(model, joint):
    """One driven link (hinge, and a spherical joint turning about one of its axes): I th'' = kp (th* - th) - kd th' with I = the
    link's inertia about the joint axis + the armature.  The drive is integrated implicitly, so every substep must be the implicit
    Euler step of that equation,  w+ = (I w + h kp (th* - th)) / (I + h kd + h^2 kp),  th+ = th + h w+  (computed here in float64
    from the physical constants - nothing of the kernel's articulated-body recursion), and the trajectory must approach the closed-
    form step response of the damped oscillator as h shrinks."""
    kp, kd, arm, target = 40.0, 3.0, 0.02, 0.5
    inertia = 0.23 + arm
    sim = make_arm(model, joint, kp, kd, arm)
    D = sim.D
    act = np.zeros((1, D), np.float32)
    act[0, 1 if joint == 2 else 0] = target                    # spherical: exp-map target about y
    col = 1 if joint == 2 else 0
    h = 1.0 / 120.0
    th, w = 0.0, 0.0
    for step in range(60):
        sim.step(act, n_sub=4, h=h)
        for _ in range(4):
            w = (inertia * w + h * kp * (target - th)) / (inertia + h * kd + h * h * kp)
            th += h * w
        assert abs(sim.dof_state[0, col, 0] - th) < 5e-5, (step, sim.dof_state[0, col, 0], th)
        assert abs(sim.dof_state[0, col, 1] - w) < 5e-4, (step, sim.dof_state[0, col, 1], w)
        if joint == 2:
            assert np.abs(sim.dof_state[0, [0, 2], :]).max() < 1e-5         # the other two axes stay put
    assert np.abs(sim.root_state[0, 7:13]).max() < 1e-4                      # the base does not move
    # closed form of the under-damped oscillator (zeta = 0.47): first-order convergence in h
    wn = np.sqrt(kp / inertia)
    zeta = kd / (2.0 * np.sqrt(kp * inertia))
    wd = wn * np.sqrt(1.0 - zeta * zeta)

    def exact(t):
        return target * (1.0 - np.exp(-zeta * wn * t) * (np.cos(wd * t) + zeta / np.sqrt(1.0 - zeta * zeta) * np.sin(wd * t)))
    errs = []
    for n_sub in (4, 32):
        sim = make_arm(model, joint, kp, kd, arm)
        worst = 0.0
        for step in range(30):
            sim.step(act, n_sub=n_sub, h=1.0 / (30.0 * n_sub))
            worst = max(worst, abs(float(sim.dof_state[0, col, 0]) - exact((step + 1) / 30.0)))
        errs.append(worst)
    assert errs[0] < 0.06 * target and errs[1] < 0.01 * target and errs[1] < 0.2 * errs[0], errs


def test_joint_drive_torque_limit(model):
    """A drive that asks for more than its effort limit pushes with the limit: far from the target the link accelerates at
    limit / I (to O(h): the implicit terms are scaled down with the torque), and the torque is the full PD law again once the error is
    small enough."""
    kp, kd, arm, lim = 400.0, 3.0, 0.02, 5.0
    inertia = 0.23 + arm
    sim = make_arm(model, 1, kp, kd, arm, effort=lim)
    act = np.full((1, 1), 2.0, np.float32)                      # kp * 2 rad = 800 N m asked for, 5 N m allowed
    h = 1.0 / 120.0
    sim.step(act, n_sub=4, h=h)
    acc = sim.dof_state[0, 0, 1] / (4 * h)
    assert 0.9 * lim / inertia < acc <= lim / inertia + 1e-4, (acc, lim / inertia)
    # an unlimited drive of the same gains does not respect it (the limit is what held the first one back)
    free = make_arm(model, 1, kp, kd, arm)
    free.step(act, n_sub=4, h=h)
    assert free.dof_state[0, 0, 1] > 10.0 * sim.dof_state[0, 0, 1]
    # inside the limit (error 0.01 rad -> 4 N m) the step is the unlimited one
    small = np.full((1, 1), 0.01, np.float32)
    a, b = make_arm(model, 1, kp, kd, arm, effort=lim), make_arm(model, 1, kp, kd, arm)
    a.step(small, n_sub=4, h=h)
    b.step(small, n_sub=4, h=h)
    np.testing.assert_array_equal(a.dof_state, b.dof_state)


def test_torque_free_asymmetric_body_conserves_angular_momentum_and_energy(model):
    """A free rigid body with three different principal inertias, spun near its intermediate axis (the unstable one: it flips over
    within two seconds), no force on it: the angular momentum in WORLD axes and the kinetic energy are constants of Euler's equations
    - the only thing at work is the gyroscopic term w x I w of the root.  Semi-implicit Euler keeps both to O(h)."""
    I = np.array([0.10, 0.22, 0.40])

    def run(h, n_sub):
        sim = make_ball(model, 10.0, 0.2, gravity=0.0, hf=np.full((60, 20), -100.0, np.float32))
        for k, v in enumerate([I[0], 0.0, 0.0, I[1], 0.0, I[2]]):
            sim.m.inertia_o[0][k] = v
        sim.root_state[0, 0:3] = [0.0, 0.0, 3.0]
        # root angular velocity in world axes; the body starts aligned with them
        sim.root_state[0, 10:13] = [0.05, 6.0, 0.05]
        act = np.zeros((1, 0), np.float32)
        L, T, wy = [], [], []
        for _ in range(60):
            R = rotm(sim.root_state[0, 3:7].astype(np.float64))
            w = sim.root_state[0, 10:13].astype(np.float64)
            wb = R.T @ w
            L.append(R @ (I * wb))
            T.append(0.5 * wb @ (I * wb))
            wy.append(wb[1])
            sim.step(act, n_sub=n_sub, h=h)
        return np.array(L), np.array(T), np.array(wy)
    L1, T1, wy1 = run(1.0 / 120.0, 4)
    L4, T4, _ = run(1.0 / 480.0, 16)
    assert wy1.min() < -3.0                                              # it did flip: the spin about the body's own y axis reversed
    dL1, dL4 = np.abs(L1 - L1[0]).max(), np.abs(L4 - L4[0]).max()
    dT1, dT4 = np.abs(T1 - T1[0]).max(), np.abs(T4 - T4[0]).max()
    # (the gyroscopic term is integrated explicitly: 4 % / 8 % over two seconds of tumbling at 6 rad/s with h = 1/120 s, a quarter of
    # that with a quarter of the step)
    print("drift of |L|, T at h = 1/120:", dL1 / np.linalg.norm(L1[0]), dT1 / T1[0], "at h = 1/480:", dL4 / np.linalg.norm(L1[0]), dT4 / T1[0])
    assert dL1 < 0.06 * np.linalg.norm(L1[0]) and dT1 < 0.12 * T1[0], (dL1, dT1, L1[0], T1[0])
    assert dL4 < 0.4 * dL1 and dT4 < 0.4 * dT1, (dL1, dL4, dT1, dT4)


def test_drive_against_a_joint_limit_rests_where_the_two_springs_balance(model):
    """A drive whose target lies beyond the joint's upper limit: at rest the drive's spring and the limit's spring carry the same
    torque, kp (th* - th) = limit_kp (th - hi), so th = (kp th* + limit_kp hi) / (kp + limit_kp), with nothing moving."""
    kp, target, hi = 40.0, 0.8, 0.3
    sim = make_arm(model, 1, kp, 3.0, 0.02)
    sim.m.limit_hi[0] = hi
    lk = float(sim.m.limit_kp)
    assert lk > 10.0 * kp                                            # the limit is the stiffer spring by far
    act = np.full((1, 1), target, np.float32)
    for _ in range(90):
        sim.step(act, n_sub=4, h=1.0 / 120.0)
    rest = (kp * target + lk * hi) / (kp + lk)
    assert hi < rest < hi + 0.05
    assert abs(sim.dof_state[0, 0, 0] - rest) < 2e-5, (sim.dof_state[0, 0, 0], rest)
    assert abs(sim.dof_state[0, 0, 1]) < 1e-4


def test_two_coaxial_rotors_share_the_drive_torque(model):
    """A light base and a link whose centres of mass both sit on the hinge axis: two rotors on one shaft.  The drive is an internal
    torque, so Ia wa + Ib wb stays zero (the base turns the other way by Ib / Ia) and the RELATIVE angle is the spring-damper of the
    reduced inertia Ia Ib / (Ia + Ib) plus the armature - the implicit-Euler recurrence of test_joint_drive_... with that inertia."""
    kp, kd, arm, target = 40.0, 3.0, 0.02, 0.5
    Ia, Ib = 0.30, 0.20
    sim = make_arm(model, 1, kp, kd, arm)
    sim.m.mass[0] = 5.0
    for k, v in enumerate([Ia, 0.0, 0.0, Ia, 0.0, Ia]):
        sim.m.inertia_o[0][k] = v
    for k in range(3):
        sim.m.com[1][k] = 0.0
    for k, v in enumerate([0.05, 0.0, 0.0, Ib, 0.0, 0.01]):
        sim.m.inertia_o[1][k] = v
    inertia = Ia * Ib / (Ia + Ib) + arm
    act = np.full((1, 1), target, np.float32)
    h = 1.0 / 120.0
    th, w = 0.0, 0.0
    for step in range(45):
        sim.step(act, n_sub=4, h=h)
        for _ in range(4):
            w = (inertia * w + h * kp * (target - th)) / (inertia + h * kd + h * h * kp)
            th += h * w
        assert abs(sim.dof_state[0, 0, 0] - th) < 1e-4, (step, sim.dof_state[0, 0, 0], th)
        wa = float(sim.root_state[0, 11])                            # base spin about y (world = body axes: it only turns about y)
        wb = wa + float(sim.dof_state[0, 0, 1])
        assert abs(Ia * wa + Ib * wb) < 2e-4 * max(1.0, abs(Ib * wb)), (step, wa, wb)
    assert np.abs(sim.root_state[0, 7:10]).max() < 1e-5               # nothing pushes the pair anywhere


def test_free_fall_does_not_load_an_undriven_joint(model):
    """Gravity accelerates every body alike: an undriven link hinged to a falling base, its centre of mass 0.3 m to the side of the
    joint, feels no torque about the joint - the angle stays where it is while both fall at g (a gravity that acted on the base alone,
    or as a force at the wrong point, would swing it)."""
    sim = make_arm(model, 1, kp=0.0, kd=0.0, armature=0.0, gravity=9.81)
    sim.m.mass[0] = 20.0                                               # a base the link could drag around
    for k, v in enumerate([2.0, 0.0, 0.0, 2.0, 0.0, 2.0]):
        sim.m.inertia_o[0][k] = v
    sim.dof_state[0, 0, 0] = 0.5 * np.pi                               # link horizontal: the lever arm of its weight is the full 0.3 m
    act = np.zeros((1, 1), np.float32)
    h, steps = 1.0 / 120.0, 20
    z0 = float(sim.root_state[0, 2])
    for _ in range(steps):
        sim.step(act, n_sub=4, h=h)
    k = 4 * steps
    assert abs(sim.dof_state[0, 0, 0] - 0.5 * np.pi) < 2e-5 and abs(sim.dof_state[0, 0, 1]) < 1e-4
    assert abs(sim.root_state[0, 9] + 9.81 * k * h) < 1e-3
    assert abs(sim.root_state[0, 2] - (z0 - 9.81 * h * h * k * (k + 1) / 2)) < 1e-3
    assert np.abs(sim.root_state[0, 10:13]).max() < 1e-4               # and the base does not start to turn

