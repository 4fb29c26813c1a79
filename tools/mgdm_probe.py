"""Step rate of an env whose rows are split between dataset clips and generated plans (fraction_dm_envs < 1), with a stand-in planner
(straight walk towards the target, 45 frames): (a) env.step + env.reset(done ids) as an eager loop issues them, and the cost of a
replan (generator excluded / included); (b) the way the agent's rollout issues them since round 3: env.step + env.reset_done captured
ONCE in a hipGraph and replayed, the steps that end in a replan through the eager reset.
python tools/mgdm_probe.py [envs] [steps]   -> one JSON line per fraction"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parc_amd import workloads  # noqa: E402
from parc_amd.envs import base_env  # noqa: E402
from parc_amd.util.motion_util import MotionFrames  # noqa: E402


class WalkGenerator:
    _num_prev_states, _sequence_fps = 2, 30
    _dx = _dy = 0.4
    _num_x_neg, _num_x_pos, _num_y_neg, _num_y_pos = 2, 5, 3, 3
    F = 45
    seconds = 0.0

    def __call__(self, target_xy, prev_frames, terrain, char_model, settings):
        torch.cuda.synchronize()
        t0 = time.time()
        F = self.F
        p, q, j = prev_frames.root_pos[:, -1], prev_frames.root_rot[:, -1], prev_frames.joint_rot[:, -1]
        d = target_xy[:, 0:2] - p[:, 0:2]
        d = d / torch.linalg.vector_norm(d, dim=-1, keepdim=True).clamp(min=1e-3)
        tt = torch.arange(F, dtype=torch.float32, device=p.device).reshape(1, F, 1) / 30.0
        rp = p.unsqueeze(1).repeat(1, F, 1)
        rp[..., 0:2] += tt * d.unsqueeze(1)
        out = MotionFrames(root_pos=rp, root_rot=q.unsqueeze(1).repeat(1, F, 1), joint_rot=j.unsqueeze(1).repeat(1, F, 1, 1),
                           contacts=torch.zeros((p.shape[0], F, char_model.get_num_joints()), device=p.device))
        torch.cuda.synchronize()
        WalkGenerator.seconds += time.time() - t0
        return out


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    dev = "cuda:0"
    for fraction in (1.0, 0.5, 0.0):
        over = {"enable_replan_timer_obs": False, "fraction_dm_envs": fraction}
        if fraction < 1.0:
            over["mgdm"] = {"plan_length": 1.0, "ddim_stride": 50, "max_replans": 5, "cfg_scale": 0.7, "target_dist_max": 4.0, "target_dist_min": 1.0,
                            "target_dur_max": 2.0, "target_dur_min": 1.0, "target_heading_scale": 0.5, "generator": WalkGenerator(),
                            "heightmap": {"horizontal_scale": 0.4, "sq_m_per_env": 2.0, "safety_region": 15.0, "num_segments": 32,
                                          "platform_heights": [0.6]}}
        t0 = time.time()
        env, _, _ = workloads.build_env("boxes_64clips", N, dev, seed=2, env_overrides=over)
        torch.cuda.synchronize()
        build_s = time.time() - t0
        obs, _ = env.reset()
        low, high = env._action_bound_low, env._action_bound_high
        WalkGenerator.seconds = 0.0
        reset_s, replans, dones = 0.0, 0, 0
        mg = env.get_mgdm_env() if env.has_mgdm_envs() else None
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            a = torch.minimum(torch.maximum(env._ref_dof_pos, low), high)
            obs, r, done, info = env.step(a)
            ids = (done != base_env.DoneFlags.NULL.value).nonzero().flatten()
            dones += int(ids.numel())
            pending = mg is not None and mg._replan_flag
            if pending:
                torch.cuda.synchronize()
                t1 = time.time()
            env.reset(ids)
            if pending:
                torch.cuda.synchronize()
                reset_s += time.time() - t1
                replans += 1
        torch.cuda.synchronize()
        dt = time.time() - t0
        # ---- (b) captured step + device-side restart, replayed
        graph_ms = None
        a_buf = torch.minimum(torch.maximum(env._ref_dof_pos, low), high).clone()
        env._info_snapshots = False
        g, gsteps, greplans = None, 0, 0
        for _ in range(3):                                   # eager warm-up of the paths the capture will record
            if env.supports_device_reset():
                env.step(a_buf)
                env.reset_done()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            a_buf.copy_(torch.minimum(torch.maximum(env._ref_dof_pos, low), high))
            if env.supports_device_reset():
                if g is None:
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        env.step(a_buf)
                        env.reset_done()
                    g.replay()
                else:
                    g.replay()
                    env.host_step_replayed()
            else:                                            # this step ends in a replan: eager step + the reset that calls the planner
                _, _, done, _ = env.step(a_buf)
                env.reset((done != base_env.DoneFlags.NULL.value).nonzero().flatten())
                greplans += 1
            gsteps += 1
        torch.cuda.synchronize()
        graph_ms = 1e3 * (time.time() - t0) / gsteps
        # replayed steps alone (no replan in the window)
        pure = None
        if g is not None:
            k = 0
            torch.cuda.synchronize()
            t0 = time.time()
            while k < 200:
                if not env.supports_device_reset():
                    _, _, done, _ = env.step(a_buf)
                    env.reset((done != base_env.DoneFlags.NULL.value).nonzero().flatten())
                    torch.cuda.synchronize()
                    t0 = time.time()
                    k = 0
                    continue
                g.replay()
                env.host_step_replayed()
                k += 1
                if k == 20:
                    break
            torch.cuda.synchronize()
            pure = 1e3 * (time.time() - t0) / max(k, 1)
        print(json.dumps({"envs": N, "fraction_dm_envs": fraction, "steps": steps, "build_s": round(build_s, 2), "ms_per_step": round(1e3 * dt / steps, 3),
                          "ms_per_step_captured_incl_replans": round(graph_ms, 3), "replans_in_captured_run": greplans,
                          "ms_per_replayed_step": None if pure is None else round(pure, 3),
                          "plan_clock_device_vs_host": [float(mg._mgdm_time_buf[0]), float(mg._plan_time_host)] if mg is not None else None,
                          "env_steps_per_s": round(N * steps / dt), "episodes_ended": dones, "replans": replans,
                          "ms_per_replan_with_generator": round(1e3 * reset_s / max(replans, 1), 3),
                          "ms_per_replan_generator_alone": round(1e3 * WalkGenerator.seconds / max(replans, 1), 3),
                          "finite": bool(torch.isfinite(obs).all())}))
        del env


if __name__ == "__main__":
    main()
