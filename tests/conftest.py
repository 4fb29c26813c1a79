import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
TOOLS = os.path.join(REPO, "tools")
if TOOLS not in sys.path:
    sys.path.append(TOOLS)          # `import parc_diag`: the diagnostics library's loader (tests of the reference simulator kernel)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def ref_char(oracle):
    return oracle.Char.from_npz(golden("g2_char"))


@pytest.fixture(scope="session")
def ref_mlib(oracle, ref_char):
    z = golden("g3_motion")
    clips = [z["frames_%d" % i] for i in range(4)]
    contacts = [z["contacts_%d" % i] for i in range(4)]
    return oracle.MotionLib(ref_char, clips, z["clip_fps"], z["clip_loop"], z["clip_weights_in"], contacts)
