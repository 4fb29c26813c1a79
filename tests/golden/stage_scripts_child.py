#!/usr/bin/env python3
"""Child process of gen_golden.py stage `stage-scripts` (build container only): runs ONE of the reference's stage scripts --
parc_3_tracker.train_tracker (parc_3_tracker.py:8-78) or parc_4_phys_record.record_motions (parc_4_phys_record.py:8-65) -- UNCHANGED,
on top of this package: `parc_amd.install_reference_aliases()` provides every module they and the reference's run.py import
(envs.env_builder, learning.agent_builder, util.arg_parser / logger / mp_util / util, PARC.util.create_dataset), only the three script
files themselves come from /root/reference.  env_builder.build_env and agent_builder.build_agent are replaced by recorders (building
the env needs the GPU), so what is captured is exactly what the scripts hand over: the YAML files they write, the argv they build for
run.main, and every call run.run makes into the package (run.py:95-138).  Prints one JSON object.

usage: stage_scripts_child.py tracker|record <config.yaml> <reference root>
"""
import json
import os
import sys

import yaml

which, cfg_path, REF = sys.argv[1], sys.argv[2], sys.argv[3]
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import parc_amd  # noqa: E402

parc_amd.install_reference_aliases()
before = set(sys.modules)
sys.path.append(REF)                      # AFTER the aliases: only what the package does not provide can come from here
os.chdir(REF)                             # the default configs name PARC/tracker_config/*.yaml relative to the reference root

import envs.env_builder as env_builder  # noqa: E402
import learning.agent_builder as agent_builder  # noqa: E402
import util.mp_util as mp_util  # noqa: E402
import util.util as util_mod  # noqa: E402

calls = []


def plain(x):
    if isinstance(x, (str, bool)) or x is None:
        return x
    if isinstance(x, (StubEnv, StubAgent)):
        return x.tag
    try:
        import numpy as np
        if isinstance(x, np.integer):
            return {"int": int(x), "numpy": type(x).__name__}
    except ImportError:
        pass
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        return x
    return repr(x)


def rec(name, args, kwargs):
    calls.append({"call": name, "args": [plain(a) for a in args], "kwargs": {k: plain(v) for k, v in kwargs.items()}})


class StubEnv:
    tag = "<env>"


class StubAgent:
    tag = "<agent>"

    def load(self, *a, **k):
        rec("agent.load", a, k)

    def train_model(self, *a, **k):
        rec("agent.train_model", a, k)

    def record_motions(self, *a, **k):
        rec("agent.record_motions", a, k)

    def test_model(self, *a, **k):
        rec("agent.test_model", a, k)
        return {"mean_return": 0.0, "mean_ep_len": 0.0, "num_eps": 0}


def build_env(*a, **k):
    rec("env_builder.build_env", a, k)
    return StubEnv()


def build_agent(*a, **k):
    rec("agent_builder.build_agent", a, k)
    return StubAgent()


env_builder.build_env = build_env
agent_builder.build_agent = build_agent
_init, _seed = mp_util.init, util_mod.set_rand_seed


def init(*a, **k):
    rec("mp_util.init", a, k)
    return _init(*a, **k)


def set_rand_seed(*a, **k):
    rec("util.set_rand_seed", a, k)
    return _seed(*a, **k)


mp_util.init = init
util_mod.set_rand_seed = set_rand_seed

import run  # noqa: E402  (the reference's launcher)

argvs = []
_main = run.main


def main(argv):
    argvs.append(list(argv))
    return _main(argv)


run.main = main
with open(cfg_path) as f:
    config = yaml.safe_load(f)
if which == "tracker":
    import parc_3_tracker  # noqa: E402
    parc_3_tracker.train_tracker(config)
else:
    import parc_4_phys_record  # noqa: E402
    parc_4_phys_record.record_motions(config)
from_ref = sorted(m for m in set(sys.modules) - before if getattr(sys.modules[m], "__file__", None) and
                  os.path.abspath(sys.modules[m].__file__).startswith(os.path.abspath(REF)))
print("G24JSON " + json.dumps({"argv": argvs, "calls": calls, "modules_loaded_from_the_reference": from_ref}))
