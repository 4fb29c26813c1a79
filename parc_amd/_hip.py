"""ctypes binding of libparc_hip.so (include/parc_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
Tensors are handed over as raw device pointers (``tensor.data_ptr()``) together with the current
HIP stream of torch; nothing here allocates or synchronises.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libparc_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["parc_kin.hip", "parc_sim.hip", "parc_ppo.hip", "parc_terrain.hip"]
# The diagnostics library (tools/parc_diag.py; never loaded by this package): the same sources with -DPARC_DIAG_BUILD (timing-ablation bits of
# parc_track_post_step, the heightmap kernel's measurement knobs) plus the one-env-per-lane reference formulation of the simulator.
DIAG_LIB_PATH = os.path.join(LIB_DIR, "libparc_hip_diag.so")
DIAG_SOURCES = SOURCES + ["parc_sim_ref.hip"]

MAX_BODIES = 16
MAX_DOFS = 64
MAX_TAR_STEPS = 6
MAX_KEY_BODIES = 8

POST_REF = 1
POST_OBS = 2
POST_REWARD_DONE = 4
POST_HF = 8
POST_MASKED = 16
POST_INIT_CHAR = 32
POST_TARGETS = 64
POST_PLAN_CLOCK = 128

c_int = ctypes.c_int
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_vp = ctypes.c_void_p


class CharModelS(ctypes.Structure):
    _fields_ = [("num_bodies", c_i32), ("dof_size", c_i32), ("max_depth", c_i32),
                ("parent", c_i32 * MAX_BODIES), ("joint_type", c_i32 * MAX_BODIES), ("dof_idx", c_i32 * MAX_BODIES),
                ("depth", c_i32 * MAX_BODIES),
                ("local_translation", (c_f * 3) * MAX_BODIES), ("local_rotation", (c_f * 4) * MAX_BODIES),
                ("joint_axis", (c_f * 3) * MAX_BODIES)]


class MotionLibS(ctypes.Structure):
    _fields_ = [("num_motions", c_i32), ("num_bodies", c_i32), ("dof_size", c_i32), ("row_stride", c_i32),
                ("off_pos", c_i32), ("off_contacts", c_i32), ("off_root_vel", c_i32), ("off_root_ang_vel", c_i32),
                ("off_dof_vel", c_i32),
                ("num_frames", c_vp), ("start_idx", c_vp), ("length", c_vp), ("loop_mode", c_vp), ("pos_delta", c_vp),
                ("frames", c_vp)]


class TerrainS(ctypes.Structure):
    _fields_ = [("hf", c_vp), ("dim_x", c_i32), ("dim_y", c_i32), ("min_x", c_f), ("min_y", c_f), ("dx", c_f), ("dy", c_f)]


class TrackCfgS(ctypes.Structure):
    _fields_ = [("num_tar_steps", c_i32), ("tar_dt", c_f * MAX_TAR_STEPS),
                ("num_key_bodies", c_i32), ("key_body_ids", c_i32 * MAX_KEY_BODIES),
                ("joint_err_w", c_f * MAX_BODIES), ("dof_err_w", c_f * MAX_DOFS), ("contact_w", c_f * MAX_BODIES),
                ("reward_w", c_f * 5), ("rel_deepmimic_w", c_f),
                ("pose_termination_dist", c_f * MAX_BODIES),
                ("pose_termination", c_i32), ("enable_early_termination", c_i32), ("track_root", c_i32),
                ("root_pos_termination_dist", c_f), ("root_rot_termination_angle", c_f), ("termination_height", c_f),
                ("num_contact_bodies", c_i32), ("contact_body_mask", c_i32 * MAX_BODIES),
                ("episode_length", c_f), ("contact_eps", c_f), ("min_obs_h", c_f), ("max_obs_h", c_f),
                ("num_ray_points", c_i32), ("obs_dim", c_i32),
                ("task1_w", c_f), ("task2_w", c_f), ("target_radius", c_f),
                ("target_future_min", c_f), ("target_future_max", c_f), ("track_root_h", c_i32), ("use_contact_info", c_i32), ("global_obs", c_i32)]


class EnvBuffersS(ctypes.Structure):
    _fields_ = [("num_envs", c_i32),
                ("root_state", c_vp), ("dof_state", c_vp), ("rigid_body_state", c_vp), ("contact_forces", c_vp),
                ("env_offsets", c_vp), ("motion_ids", c_vp), ("motion_time_offsets", c_vp), ("motion_xy_offset", c_vp),
                ("time_buf", c_vp), ("target_xy", c_vp),
                ("ref_root_pos", c_vp), ("ref_root_rot", c_vp), ("ref_root_vel", c_vp), ("ref_root_ang_vel", c_vp),
                ("ref_joint_rot", c_vp), ("ref_dof_vel", c_vp), ("ref_dof_pos", c_vp),
                ("ref_contacts", c_vp), ("ref_body_pos", c_vp),
                ("obs", c_vp), ("reward", c_vp), ("reward_terms", c_vp), ("done", c_vp), ("done_kind", c_vp),
                ("env_mask", c_vp), ("init_noise_xy", c_vp), ("next_target_time", c_vp), ("target_rand", c_vp), ("obs_aux", c_vp),
                ("reward_terms_stride", c_i32)]


class RecordFieldS(ctypes.Structure):
    _fields_ = [("src", c_vp), ("dst", c_vp), ("row_bytes", c_i32), ("convert", c_i32)]


class PPOCfgS(ctypes.Structure):
    _fields_ = [("clip_ratio", c_f), ("bound_w", c_f), ("entropy_w", c_f), ("reg_w", c_f), ("critic_w", c_f),
                ("large_critic_loss", c_f), ("critic_l1", c_i32)]


_lib = None


# per-source optimisation level.  parc_sim_ref.hip (the one-env-per-lane reference kernel, diagnostics library only) is built at
# -O2: at -O3 hipcc (ROCm 7.2, gfx950) miscompiles it (results diverge from the -O0/-O1/-O2 builds and from the g++ host build of the
# same source; GVN scalar PRE on the unrolled 3x3 helpers).  The product's simulator kernels (parc_sim.hip) are correct at
# every level (profiles/r02_sim_o3_bisect.txt) and are built at whichever measured faster: -O3 WITHOUT the SLP vectoriser since round 4
# (its v_pk_* pairs come with a third more register moves: 114.9 -> 100.8 us per 4096-env step, profiles/r04_sim_step_variants.txt).
OPT_LEVEL = {"parc_kin.hip": os.environ.get("PARC_KIN_OPT", "-O3 -fno-slp-vectorize"), "parc_sim.hip": os.environ.get("PARC_SIM_OPT", "-O3 -fno-slp-vectorize"), "parc_sim_ref.hip": "-O2"}


DIGEST_PATH = os.path.join(LIB_DIR, "libparc_hip.digest")
DIAG_DIGEST_PATH = os.path.join(LIB_DIR, "libparc_hip_diag.digest")


def source_digest():
    """sha256 over every source the library is built from (csrc/*, include/*.h) and the per-source flags.  Written next to the library
    by build(); lib() refuses a library whose digest differs from the sources it sits beside - a stale .so would be called with
    argument structs of another layout (file times are not used: a snapshot copy does not keep them)."""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + \
        ([os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")] if os.path.isdir(inc) else [])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(repr((DIAG_SOURCES, sorted(OPT_LEVEL.items()))).encode())
    return h.hexdigest()


def _stored_digest(path=DIGEST_PATH):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def build(force=False, verbose=False, diag=False):
    """Compile the HIP sources for gfx950 into parc_amd/lib/libparc_hip.so (hipcc cross-compiles without a GPU).
    diag=True: the diagnostics library libparc_hip_diag.so instead (tools/parc_diag.py is its only user)."""
    lib_path, digest_path, sources = (DIAG_LIB_PATH, DIAG_DIGEST_PATH, DIAG_SOURCES) if diag else (LIB_PATH, DIGEST_PATH, SOURCES)
    digest = source_digest()
    if not force and os.path.exists(lib_path) and _stored_digest(digest_path) == digest:
        return lib_path
    obj_dir = os.path.join(LIB_DIR, "obj_diag" if diag else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for s in sources:
        o = os.path.join(obj_dir, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950"] + OPT_LEVEL.get(s, "-O3").split() + (["-DPARC_DIAG_BUILD"] if diag else []) + \
            ["-std=c++17", "-fPIC", "-c", "-o", o, os.path.join(CSRC, s)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))      # the translation units are independent: compile them side by side
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(digest_path, "w") as f:
        f.write(digest + "\n")
    return lib_path


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libparc_hip.so is not built ({}); run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "-- there is no CPU fallback".format(LIB_PATH))
        if _stored_digest() != source_digest():
            raise RuntimeError("libparc_hip.so was built from other sources than the ones beside it (csrc/, include/): rebuild with "
                               "`python -c 'import __graft_entry__ as g; g.build()'` - calling a stale library would hand it argument "
                               "structs of another layout")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    L.parc_abi_version.restype = c_int
    L.parc_refresh_ray_obs_hfs.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_vp, TerrainS, c_f, c_f, c_vp, c_i64]
    L.parc_refresh_obs_hfs.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_vp, TerrainS, c_f, c_f, c_vp, c_i64]
    L.parc_dof_to_rot.argtypes = [c_vp, CharModelS, c_int, c_vp, c_vp]
    L.parc_rot_to_dof.argtypes = [c_vp, CharModelS, c_int, c_vp, c_vp]
    L.parc_forward_kinematics.argtypes = [c_vp, CharModelS, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_pose_chain_forward.argtypes = [c_vp, CharModelS, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_pose_chain_backward.argtypes = [c_vp, CharModelS, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_body_points_world.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_body_points_world_grad.argtypes = [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_quat_diff_angle.argtypes = [c_vp, ctypes.c_int64, c_vp, c_vp, c_vp]
    L.parc_quat_diff_angle_grad.argtypes = [c_vp, ctypes.c_int64, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_temporal_terms.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_f, c_f, c_vp]
    L.parc_temporal_terms_grad.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_f, c_f, c_vp, c_vp, c_vp]
    L.parc_calc_motion_frame.argtypes = [c_vp, MotionLibS, c_int, c_vp, c_vp] + [c_vp] * 7
    L.parc_motion_lib_build.argtypes = [c_vp, CharModelS, MotionLibS, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_track_post_step.argtypes = [c_vp, CharModelS, MotionLibS, TerrainS, TrackCfgS, EnvBuffersS, c_vp, c_int, c_int, c_vp]
    L.parc_track_post_step_timed.argtypes = L.parc_track_post_step.argtypes + [c_vp, c_vp]
    L.parc_track_post_step_timed.restype = c_int
    L.parc_assemble_obs.argtypes = [c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int]
    L.parc_assemble_obs.restype = c_int
    L.parc_update_fail_rates.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_f, c_vp]
    L.parc_step_tail.argtypes = [c_vp, CharModelS, MotionLibS, EnvBuffersS, c_int, c_int, c_vp, c_f, c_vp]
    L.parc_step_tail.restype = c_int
    L.parc_td_lambda_return.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_f, c_f, c_vp]
    L.parc_adv_normalize.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_vp]
    L.parc_ppo_loss.argtypes = [c_vp, c_int, c_int] + [c_vp] * 8 + [PPOCfgS] + [c_vp] * 5
    L.parc_ppo_loss.restype = c_int
    L.parc_ppo_loss_packed.argtypes = [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, PPOCfgS] + [c_vp] * 5
    L.parc_ppo_loss_packed.restype = c_int
    L.parc_ppo_workspace_floats.argtypes = [c_int]
    L.parc_ppo_workspace_floats.restype = c_int
    L.parc_normalize_clamp.argtypes = [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_f, c_vp]
    L.parc_normalize_clamp.restype = c_int
    L.parc_action_head.argtypes = [c_vp, c_int, c_int] + [c_vp] * 8
    L.parc_action_head.restype = c_int
    L.parc_action_head_record.argtypes = [c_vp, c_int, c_int] + [c_vp] * 8 + [c_vp] * 5 + [c_int, c_vp]
    L.parc_action_head_record.restype = c_int
    L.parc_points_hf_sdf.argtypes = [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 5 + [c_f, c_f, c_f, c_int, c_f, c_vp, c_vp]
    L.parc_points_hf_sdf.restype = c_int
    L.parc_points_hf_sdf_grad.argtypes = [c_vp, c_int, c_int, c_int, c_int] + [c_vp] * 5 + [c_f, c_f, c_f, c_int, c_vp, c_vp, c_vp]
    L.parc_points_hf_sdf_grad.restype = c_int
    L.parc_scale_by_clipped_norm.argtypes = [c_vp, c_i64, c_vp, c_vp, c_f]
    L.parc_scale_by_clipped_norm.restype = c_int
    L.parc_moments_workspace_floats.argtypes = [c_i64, c_int]
    L.parc_moments_workspace_floats.restype = c_i64
    L.parc_moments_accumulate.argtypes = [c_vp, c_i64, c_int, c_vp, c_vp, c_vp]
    L.parc_moments_accumulate.restype = c_int
    L.parc_relu_bwd_workspace_floats.argtypes = [c_i64, c_int]
    L.parc_relu_bwd_workspace_floats.restype = c_i64
    L.parc_relu_bwd_bias_grad.argtypes = [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]
    L.parc_relu_bwd_bias_grad.restype = c_int
    L.parc_weighted_colsum.argtypes = [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp]
    L.parc_weighted_colsum.restype = c_int
    L.parc_sgd_workspace_floats.argtypes = []
    L.parc_sgd_workspace_floats.restype = c_i64
    L.parc_sgd_momentum_step.argtypes = [c_vp, c_i64, c_vp, c_vp, c_vp, c_f, c_f, c_f, c_f, c_vp, c_vp]
    L.parc_sgd_momentum_step.restype = c_int
    L.parc_return_tracker_update.argtypes = [c_vp, c_int, c_int, c_vp, c_i64] + [c_vp] * 8
    L.parc_return_tracker_workspace_floats.argtypes = [c_int]
    L.parc_return_tracker_workspace_floats.restype = c_i64
    L.parc_return_tracker_update.restype = c_int
    L.parc_record_step.argtypes = [c_vp, c_int, c_vp, c_int, c_vp]
    L.parc_record_step.restype = c_int
    L.parc_obs_ingest.argtypes = [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_vp, c_vp, c_vp]
    L.parc_obs_ingest.restype = c_int
    L.parc_rng_step.argtypes = [c_vp, ctypes.c_uint64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_int]
    L.parc_rng_step.restype = c_int
    L.parc_reset_apply.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int] + [c_vp] * 9
    L.parc_reset_apply.restype = c_int
    L.parc_reset_sample_apply.argtypes = [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_f, c_vp, c_vp, c_int, c_f] + [c_vp] * 11
    L.parc_reset_sample_apply.restype = c_int
    for name in ("parc_refresh_ray_obs_hfs", "parc_refresh_obs_hfs", "parc_dof_to_rot", "parc_rot_to_dof",
                 "parc_forward_kinematics", "parc_calc_motion_frame", "parc_motion_lib_build", "parc_track_post_step", "parc_reset_apply",
                 "parc_update_fail_rates", "parc_td_lambda_return", "parc_adv_normalize"):
        getattr(L, name).restype = c_int
    if hasattr(L, "parc_sim_abi"):
        from . import _hip_sim
        _hip_sim.declare(L)


EXPORTED = ["parc_abi_version", "parc_refresh_ray_obs_hfs", "parc_refresh_obs_hfs", "parc_dof_to_rot", "parc_rot_to_dof",
            "parc_forward_kinematics", "parc_calc_motion_frame", "parc_motion_lib_build", "parc_track_post_step",
            "parc_update_fail_rates", "parc_td_lambda_return", "parc_adv_normalize", "parc_reset_apply", "parc_ppo_loss", "parc_ppo_workspace_floats", "parc_record_step", "parc_return_tracker_update", "parc_normalize_clamp",
            "parc_action_head", "parc_points_hf_sdf", "parc_moments_workspace_floats", "parc_moments_accumulate", "parc_reset_sample_apply", "parc_return_tracker_workspace_floats", "parc_scale_by_clipped_norm", "parc_relu_bwd_workspace_floats",
            "parc_relu_bwd_bias_grad", "parc_ppo_loss_packed", "parc_weighted_colsum", "parc_sgd_workspace_floats", "parc_sgd_momentum_step",
            "parc_pose_chain_forward", "parc_pose_chain_backward", "parc_points_hf_sdf_grad", "parc_body_points_world", "parc_body_points_world_grad",
            "parc_quat_diff_angle", "parc_quat_diff_angle_grad", "parc_temporal_terms", "parc_temporal_terms_grad", "parc_step_tail",
            "parc_assemble_obs", "parc_track_post_step_timed", "parc_rng_step", "parc_obs_ingest", "parc_action_head_record"]


def check(rc, what):
    if rc != 0:
        raise RuntimeError("{} failed with code {} ({})".format(what, rc, "bad argument" if rc == -1 else "unsupported" if rc == -2 else "hipError"))


def stream():
    return c_vp(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return c_vp(0)
    assert t.is_cuda, "HIP kernels need device tensors (no CPU fallback)"
    assert t.is_contiguous(), "tensor must be contiguous"
    return c_vp(t.data_ptr())


def terrain_struct(hf, min_point, dxdy):
    assert hf.dtype == torch.float32 and hf.is_contiguous() and hf.dim() == 2
    return TerrainS(ptr(hf), int(hf.shape[0]), int(hf.shape[1]), float(min_point[0]), float(min_point[1]),
                    float(dxdy[0]), float(dxdy[1]))


class HipEventPair:
    """Two raw hipEvent_t for parc_track_post_step_timed (torch creates its event handles lazily, at the first record)."""
    _rt = None

    def __init__(self):
        if HipEventPair._rt is None:
            HipEventPair._rt = ctypes.CDLL("libamdhip64.so")
        self.start, self.stop = c_vp(), c_vp()
        for ev in (self.start, self.stop):
            rc = HipEventPair._rt.hipEventCreate(ctypes.byref(ev))
            if rc != 0:
                raise RuntimeError("hipEventCreate -> %d" % rc)

    def elapsed_us(self):
        ms = ctypes.c_float()
        rc = HipEventPair._rt.hipEventElapsedTime(ctypes.byref(ms), self.start, self.stop)
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime -> %d" % rc)
        return ms.value * 1e3

    def __del__(self):
        try:
            for ev in (self.start, self.stop):
                if ev:
                    HipEventPair._rt.hipEventDestroy(ev)
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass
