"""MI355X-native hot path of the PARC motion tracker (parc_3_tracker / parc_4_phys_record).

``install_reference_aliases()`` registers this package's modules under the names the reference's scripts
import (``envs.*``, ``learning.*``, ``anim.*``, ``util.*``, ``PARC.util.create_dataset``) so that ``run.py``,
``parc_3_tracker.py`` and ``parc_4_phys_record.py`` drop in unchanged.
"""
import importlib
import sys
import types

__version__ = "0.2.0"

# reference module name -> module of this package that implements it.  The reference splits its agent over
# base_agent / ppo_agent / dm_ppo_agent; here it is one class, so the three names map to one module (which exports
# BaseAgent / PPOAgent / DMPPOAgent / AgentMode).  Procedural terrain generators live in terrain_procgen but are
# re-exported by terrain_util like in the reference.
_ALIASES = {
    "anim": "anim", "anim.kin_char_model": "anim.kin_char_model", "anim.motion_lib": "anim.motion_lib",
    "util": "util", "util.terrain_util": "util.terrain_util", "util.geom_util": "util.geom_util", "util.torch_util": "util.torch_util",
    "util.mp_util": "util.mp_util", "util.logger": "util.logger", "util.arg_parser": "util.arg_parser", "util.util": "util.util",
    "util.safe_pickle": "util.safe_pickle", "util.motion_util": "util.motion_util",
    "envs": "envs", "envs.base_env": "envs.base_env", "envs.env_builder": "envs.env_builder", "envs.ig_parkour": "envs.ig_parkour",
    "envs.ig_parkour.ig_parkour_env": "envs.ig_parkour.ig_parkour_env", "envs.ig_parkour.dm_env": "envs.ig_parkour.dm_env",
    "envs.ig_parkour.mgdm_env": "envs.ig_parkour.mgdm_env",
    "learning": "learning", "learning.rl_util": "learning.rl_util", "learning.experience_buffer": "learning.experience_buffer",
    "learning.normalizer": "learning.normalizer", "learning.agent_builder": "learning.agent_builder",
    "learning.base_agent": "learning.dm_ppo_agent", "learning.ppo_agent": "learning.dm_ppo_agent",
    "learning.dm_ppo_agent": "learning.dm_ppo_agent", "learning.mp_optimizer": "learning.mp_optimizer",
    "learning.dm_ppo_model": "learning.dm_ppo_model", "learning.dm_ppo_return_tracker": "learning.dm_ppo_return_tracker",
    "learning.tracking_error_tracker": "learning.tracking_error_tracker",
    "PARC.util.create_dataset": "util.create_dataset",
    "tools.motion_opt.motion_optimization": "tools.motion_opt.motion_optimization",
    "zmotion_editing_tools": "zmotion_editing_tools", "zmotion_editing_tools.motion_edit_lib": "zmotion_editing_tools.motion_edit_lib",
}
# pure namespace packages of the reference that hold nothing this path needs besides the sub-module above
_NAMESPACES = ("PARC", "PARC.util", "tools", "tools.motion_opt")


def install_reference_aliases(strict=True):
    """Make ``import envs.env_builder`` etc. resolve to parc_amd's implementations.

    A name that cannot be provided raises (``strict=False`` skips it instead); a name that something else already
    registered (the real reference on ``sys.path``) is left alone."""
    for ns in _NAMESPACES:
        if ns not in sys.modules:
            pkg = types.ModuleType(ns)
            pkg.__path__ = []
            sys.modules[ns] = pkg
    for name, target in _ALIASES.items():
        try:
            mod = importlib.import_module("parc_amd." + target)
        except ModuleNotFoundError:
            if strict:
                raise
            continue
        mod = sys.modules.setdefault(name, mod)
        parent, _, leaf = name.rpartition(".")
        if parent in sys.modules and not hasattr(sys.modules[parent], leaf):
            setattr(sys.modules[parent], leaf, mod)
