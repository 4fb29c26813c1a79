"""build_env(env_file, num_envs, device, visualize) -- mirror of the reference's envs/env_builder.py:6-25."""
import yaml

from ..util import mp_util
from .ig_parkour import ig_parkour_env


def load_env_file(file):
    with open(file, "r") as stream:
        return yaml.safe_load(stream)


def build_env(env_file, num_envs, device, visualize):
    env_config = env_file if isinstance(env_file, dict) else load_env_file(env_file)
    env_name = env_config["env_name"]
    device = mp_util.resolve_device(device)          # one rank per GPU under the reference's launcher (mp_util.rank_device)
    print("Building {} env".format(env_name))
    if env_name == ig_parkour_env.IGParkourEnv.NAME:
        return ig_parkour_env.IGParkourEnv(config=env_config, num_envs=num_envs, device=device, visualize=visualize)
    raise AssertionError("Unsupported env: {}".format(env_name))
