"""MI355X-native hot path of the PARC motion tracker (parc_3_tracker / parc_4_phys_record).

``install_reference_aliases()`` registers this package's sub-packages under the module names the
reference's scripts import (``envs``, ``learning``, ``anim``, ``util``) so that ``run.py``,
``parc_3_tracker.py`` and ``parc_4_phys_record.py`` drop in unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"

_ALIASES = ["anim", "anim.kin_char_model", "anim.motion_lib", "util", "util.terrain_util", "util.geom_util",
            "util.safe_pickle", "envs", "envs.base_env", "envs.env_builder", "envs.ig_parkour",
            "envs.ig_parkour.ig_parkour_env", "envs.ig_parkour.dm_env", "learning", "learning.rl_util",
            "learning.experience_buffer", "learning.normalizer", "learning.agent_builder", "learning.base_agent",
            "learning.ppo_agent", "learning.dm_ppo_agent", "learning.mp_optimizer", "learning.dm_ppo_model",
            "learning.dm_ppo_return_tracker", "util.mp_util", "util.logger", "util.torch_util", "util.arg_parser",
            "util.util"]


def install_reference_aliases(strict=False):
    """Make ``import envs.env_builder`` etc. resolve to parc_amd's implementations."""
    for name in _ALIASES:
        try:
            mod = importlib.import_module("parc_amd." + name)
        except ModuleNotFoundError:
            if strict:
                raise
            continue
        sys.modules.setdefault(name, mod)
