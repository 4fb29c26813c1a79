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


def _plain_dump(obj, path):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def build(root, make_terrain, files=None, dump=_plain_dump):
    """dump(obj, path): the writer - plain pickle.dump where the terrain class is importable under the path it names (the reference's
    own class in gen_golden.py), parc_amd's terrain_util.dump_reference_pickle for this package's SubTerrain"""
    rng = np.random.default_rng(4)
    for rel, nf, fps, dims, extra in (FILES if files is None else files):
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        d = {"frames": rng.standard_normal((nf, 34)).astype(np.float32), "fps": fps, "loop_mode": "CLAMP",
             "terrain": make_terrain(np.zeros(dims, np.float32))}
        d.update(extra)
        dump(d, p)
    return [os.path.join(root, "A"), os.path.join(root, "B")]


# ---- a small dataset of REAL motion (pieces of the two clips the reference ships, read from the committed fixtures) for the stage-script
# fixture G24: gen_golden.py builds it in the build container, the GPU test rebuilds the identical tree on the box
CLIP_FILES = [
    ("A/running/civ_a.pkl", 0, 0, 70),         # (relative path, which shipped clip, first frame, last frame)
    ("A/running/civ_b.pkl", 0, 70, 150),
    ("B/jumping/civ_c.pkl", 0, 150, 254),
    ("B/jumping/teaser.pkl", 1, 0, 58),
]


def build_clip_tree(root, make_terrain, golden_dir=None, dump=_plain_dump):
    """-> the folder list for create_dataset.  Every piece keeps its clip's own heightfield (positions stay consistent with it)."""
    golden_dir = golden_dir or os.path.dirname(os.path.abspath(__file__))
    z3 = np.load(os.path.join(golden_dir, "g3_motion.npz"))
    z5 = [np.load(os.path.join(golden_dir, n + ".npz")) for n in ("g5_hf_civ", "g5_hf_teaser")]
    for rel, which, f0, f1 in CLIP_FILES:
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        ter = make_terrain(z5[which]["hf"].astype(np.float32), z5[which]["min_point"].astype(np.float32), z5[which]["dxdy"].astype(np.float32))
        d = {"frames": z3["frames_%d" % which][f0:f1].astype(np.float32), "contacts": z3["contacts_%d" % which][f0:f1].astype(np.float32),
             "fps": 30, "loop_mode": "CLAMP", "terrain": ter}
        dump(d, p)
    return [os.path.join(root, "A"), os.path.join(root, "B")]
