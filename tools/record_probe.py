#!/usr/bin/env python3
"""Per-step cost of motion recording (IGParkourEnv.write_agent_states): the device-side recorder against the reference's scheme
(state of every env copied to the host and appended to per-env Python lists on every step).  python3 tools/record_probe.py [envs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from parc_amd import workloads
from parc_amd.util import mp_util

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
mp_util.init(0, 1, dev)
env, _, _ = workloads.build_env("boxes_64clips", N, dev, seed=0)
env._output_motion_dir = "/tmp/parc_record_probe"
env.reset()
env.build_agent_states_dict("_probe", record_obs=True)
env._done_buf.zero_()
env.write_agent_states()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    env._done_buf.zero_()
    env.write_agent_states()
torch.cuda.synchronize()
new_ms = (time.time() - t0) / 20 * 1e3

lists = [{"frames": [], "contacts": [], "obs": []} for _ in range(N)]
t0 = time.time()
for _ in range(5):
    frames, contacts = env._get_char_state_all()
    frames, contacts, obs = frames.cpu().numpy(), contacts.cpu().numpy(), env._obs_buf.cpu().numpy()
    done = env._done_buf.cpu().numpy()
    for e in range(N):
        lists[e]["frames"].append(frames[e].copy())
        lists[e]["contacts"].append(contacts[e].copy())
        lists[e]["obs"].append(obs[e].copy())
        if done[e] == 1:
            pass
old_ms = (time.time() - t0) / 5 * 1e3
print("envs %d: recorder %.2f ms / step (device buffers, one host read), per-env host lists %.1f ms / step" % (N, new_ms, old_ms))

# the record-mode loop of DMPPOAgent.record_motions (deterministic policy -> step -> record -> reset of finished envs), episodes kept
# alive so that no file is written inside the timed region
agent = workloads.build_agent(env, dev, mp_scale_rollout=False)
agent.eval()
from parc_amd.learning.dm_ppo_agent import AgentMode
agent.set_mode(AgentMode.TEST)
env._never_done = True
agent._curr_obs, agent._curr_info = env.reset()
env.build_agent_states_dict("_probe", record_obs=True)
env.write_agent_states()
for timed in (False, True):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(30):
        action, _ = agent._decide_action(agent._curr_obs, agent._curr_info)
        _, _, done, _ = env.step(action)
        agent._curr_obs, agent._curr_info = agent._reset_done_envs(done)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 30
print("record-mode loop: %.2f ms / step = %.2f M env-steps/s (policy + simulator + observation + recording + reset check)" % (dt * 1e3, N / dt / 1e6))
