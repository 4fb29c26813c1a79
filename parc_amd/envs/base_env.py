"""What an environment must offer to the agents of this package.

Interface contract of the reference's envs/base_env.py:8-69 (same names and meanings, so agents and scripts written against
the reference keep working): the two enums below and an environment object with ``reset / step`` plus a handful of
getters whose defaults describe a single, non-vectorised env with unbounded rewards.
"""
import enum
import math

# how an env is being driven (TRAIN: exploration noise, curriculum statistics; TEST: deterministic evaluation)
EnvMode = enum.Enum("EnvMode", {"TRAIN": 0, "TEST": 1})

# per-env episode status written into done_buf: still running, terminated by failure, by success, or by the time limit
DoneFlags = enum.Enum("DoneFlags", {"NULL": 0, "FAIL": 1, "SUCC": 2, "TIME": 3})


class BaseEnv:
    """Defaults for the optional queries; ``reset`` and ``step`` have to be provided by the concrete env."""

    _REQUIRED = ("reset", "step")

    def __init__(self, visualize):
        for name in self._REQUIRED:
            if getattr(type(self), name) is getattr(BaseEnv, name):
                raise TypeError("{} does not implement {}()".format(type(self).__name__, name))
        self._visualize = visualize
        self._mode = EnvMode.TRAIN
        self._action_space = None

    # -- to be implemented -------------------------------------------------------------------------------------------
    def reset(self, env_ids=None):
        """-> (obs, info); env_ids None = every env, an empty tensor = none."""
        raise NotImplementedError

    def step(self, action):
        """-> (obs, reward, done, info)"""
        raise NotImplementedError

    # -- queries with defaults -------------------------------------------------------------------------------------------
    def get_num_envs(self):
        return 1

    def get_action_space(self):
        return self._action_space

    def get_visualize(self):
        return self._visualize

    def set_mode(self, mode):
        self._mode = mode

    def get_reward_bounds(self):
        return (-math.inf, math.inf)

    def get_reward_fail(self):
        """reward-to-go assigned to a FAIL terminal state"""
        return 0.0

    def get_reward_succ(self):
        """reward-to-go assigned to a SUCC terminal state"""
        return 0.0

    def get_extra_log_info(self):
        return None

    def post_test_update(self):
        return None
