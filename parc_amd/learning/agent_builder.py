"""build_agent(agent_file, env, device) -- mirror of the reference's learning/agent_builder.py:14-37."""
import yaml

from ..util import mp_util
from . import dm_ppo_agent


def load_agent_file(file):
    with open(file, "r") as stream:
        return yaml.safe_load(stream)


def build_agent(agent_file, env, device):
    cfg = agent_file if isinstance(agent_file, dict) else load_agent_file(agent_file)
    name = cfg["agent_name"]
    device = mp_util.resolve_device(device)          # one rank per GPU under the reference's launcher (mp_util.rank_device)
    print("Building {} agent".format(name))
    if name in (dm_ppo_agent.DMPPOAgent.NAME, "PPO"):
        agent = dm_ppo_agent.DMPPOAgent(config=cfg, env=env, device=device)
    else:
        raise AssertionError("Unsupported agent: {}".format(name))
    print("Total parameter count: {}".format(agent.calc_num_params()))
    return agent
