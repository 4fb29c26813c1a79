"""Small motion-folder tree shared by gen_golden.py (which runs the reference's create_dataset on it) and the test (which runs
parc_amd.util.create_dataset on an identical tree).  ``make_terrain(hf)`` builds the terrain object stored in each file."""
import os
import pickle

import numpy as np

# (relative path, frames, fps, terrain dims, extra)
FILES = [
    ("A/running/r0.pkl", 40, 30, (10, 12), {}),
    ("A/running/r1.pkl", 65, 30, (10, 12), {}),
    ("A/running/r2.pkl", 30, 60, (10, 12), {}),
    ("A/running/r3.pkl", 90, 30, (10, 12), {}),
    ("A/running/sub/r4.pkl", 55, 30, (8, 8), {}),
    ("A/jumping/j0.pkl", 120, 30, (20, 20), {}),
    ("A/jumping/j1.pkl", 45, 30, (50, 20), {}),           # terrain too large -> excluded
    ("A/jumping/j3.pkl", 77, 30, (20, 20), {"loss": 3.0}),
    ("A/ignore_these/x0.pkl", 50, 30, (10, 10), {}),
    ("B/stairs/s0.pkl", 61, 30, (16, 16), {}),
    ("B/stairs/s1.pkl", 200, 30, (16, 16), {}),
]


# a clip whose stored generation loss is > 20: the reference means to skip it but raises TypeError in its log line
# (create_dataset.py:134 subscripts a MotionData); not part of the golden tree, the test adds it separately
BAD_LOSS_FILE = ("A/jumping/j2.pkl", 33, 30, (20, 20), {"loss": 25.0})


def build(root, make_terrain, files=None):
    rng = np.random.default_rng(4)
    for rel, nf, fps, dims, extra in (FILES if files is None else files):
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        d = {"frames": rng.standard_normal((nf, 34)).astype(np.float32), "fps": fps, "loop_mode": "CLAMP",
             "terrain": make_terrain(np.zeros(dims, np.float32))}
        d.update(extra)
        with open(p, "wb") as f:
            pickle.dump(d, f)
    return [os.path.join(root, "A"), os.path.join(root, "B")]
