"""Seeded synthetic workloads for tests, smoke and bench (there is no dataset download: the PARC iter-0 data
sits behind a login, README.md:35-36 of the reference).

Clips are smooth procedural humanoid motions in the PARC motion format ([F, 6+28] frames = root position,
root exponential map, joint dofs; [F, 15] contact labels; 30 fps).  Terrains follow the reference's procgen
recipe for boxes (util/terrain_util.py:864-917 add_boxes_to_hf2 with the parameters of
parc_2_kin_gen.py:36-43): random oriented rectangles stamped with a constant height into a 0.4 m grid.  A
corridor under the clip's root path is levelled so the reference motion is physically walkable.
"""
import numpy as np


def box_heightfield(rng, dim_x=16, dim_y=16, num_boxes=10, min_h=-3.0, max_h=3.0, min_len=5.0, max_len=10.0):
    """hf[dim_x, dim_y] with `num_boxes` random oriented boxes (cell units), later boxes overwrite earlier ones."""
    hf = np.zeros((dim_x, dim_y), dtype=np.float32)
    ii, jj = np.meshgrid(np.arange(dim_x, dtype=np.float32), np.arange(dim_y, dtype=np.float32), indexing="ij")
    for _ in range(num_boxes):
        cx, cy = rng.random() * dim_x, rng.random() * dim_y
        lx, ly = rng.random(2) * (max_len - min_len) + min_len
        ang = rng.random() * 2.0 * np.pi
        c, s = np.cos(ang), np.sin(ang)
        x, y = ii - cx, jj - cy
        rx, ry = x * c - y * s, x * s + y * c
        inside = (np.abs(rx) < lx / 2) & (np.abs(ry) < ly / 2)
        hf[inside] = np.float32(rng.random() * (max_h - min_h) + min_h)
    return hf


def walking_clip(rng, num_frames, fps=30.0, speed=1.0, start_xy=(0.0, 0.0), heading=0.0, base_height=0.0):
    """A smooth gait-like clip: the root translates along `heading`, hips/knees/shoulders swing in anti-phase."""
    t = np.arange(num_frames, dtype=np.float64) / fps
    fr = np.zeros((num_frames, 34), dtype=np.float64)
    d = np.array([np.cos(heading), np.sin(heading)])
    fr[:, 0] = start_xy[0] + d[0] * speed * t
    fr[:, 1] = start_xy[1] + d[1] * speed * t
    fr[:, 2] = base_height + 0.90 + 0.015 * np.sin(2 * np.pi * 2.0 * t)
    fr[:, 5] = heading                                   # root exp map: yaw only
    w = 2 * np.pi * (0.8 + 0.4 * rng.random())
    ph = rng.random() * 2 * np.pi
    amp = 0.25 + 0.15 * rng.random()
    # dof layout (kin tree order): abdomen 0:3, neck 3:6, r_shoulder 6:9, r_elbow 9, l_shoulder 10:13, l_elbow 13,
    # r_hip 14:17, r_knee 17, r_ankle 18:21, l_hip 21:24, l_knee 24, l_ankle 25:28  (+6 for the root columns)
    sw = np.sin(w * t + ph)
    fr[:, 6 + 15] = -amp * sw                            # right hip y (flexion)
    fr[:, 6 + 22] = amp * sw                             # left hip y
    fr[:, 6 + 17] = 0.35 * amp * (1 + np.sin(w * t + ph + 0.6)) + 0.05     # right knee
    fr[:, 6 + 24] = 0.35 * amp * (1 - np.sin(w * t + ph + 0.6)) + 0.05     # left knee
    fr[:, 6 + 7] = 0.4 * amp * sw                        # right shoulder y
    fr[:, 6 + 11] = -0.4 * amp * sw                      # left shoulder y
    fr[:, 6 + 9] = 0.4                                   # elbows slightly bent
    fr[:, 6 + 13] = -0.4
    fr[:, 6 + 6] = 1.2                                   # arms down (shoulder x)
    fr[:, 6 + 10] = -1.2
    fr[:, 6 + 1] = 0.03 * np.sin(w * t + ph)             # abdomen sway
    con = np.zeros((num_frames, 15), dtype=np.float32)
    con[:, 11] = (sw > 0.0)                              # right foot planted while the left swings
    con[:, 14] = (sw <= 0.0)
    return fr.astype(np.float32), con


def parkour_heightfield(kind, seed, dim, dx):
    """Stairs / curvy raised paths / both, from the reference's generators (util/terrain_procgen.py; BASELINE configs[4])."""
    import random
    import torch
    from .util import terrain_util
    state = (torch.random.get_rng_state(), random.getstate(), np.random.get_state())
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    try:
        t = terrain_util.SubTerrain("t", dim, dim, dx, dx, -dim * dx / 2 + dx / 2, -dim * dx / 2 + dx / 2, device="cpu")
        if kind in ("paths", "mix"):
            terrain_util.gen_paths_hf(t, num_paths=4, maxpool_size=1, floor_height=0.0, path_min_height=-0.5, path_max_height=1.5, num_points=300)
        if kind in ("stairs", "mix"):
            terrain_util.add_stairs_to_hf(t, min_stair_start_height=-1.0, max_stair_start_height=0.5, num_stairs=2)
        return t.hf.numpy().astype(np.float32).copy()
    finally:
        torch.random.set_rng_state(state[0])
        random.setstate(state[1])
        np.random.set_state(state[2])


def make_dataset(num_clips=64, seed=0, tile_cells=16, dx=0.4, frames_range=(120, 254), flat=False, boxes=10, terrain_kind="boxes",
                 tile_cells_range=None):
    """-> list of dict(frames, contacts, fps, loop, weight, hf, min_point, dxdy) with per-clip local terrains.
    terrain_kind: "boxes" (kin-gen recipe) or "stairs" / "paths" / "mix" (parkour terrains), cycled per clip for "parkour".
    tile_cells_range=(lo, hi): every clip gets its own rectangular terrain with both sides drawn from [lo, hi] cells (the iter-0
    dataset keeps terrains up to 45 x 45 cells, PARC/create_dataset_config.yaml:16-17); box terrains only."""
    rng = np.random.default_rng(seed)
    clips = []
    for k in range(num_clips):
        nf = int(rng.integers(frames_range[0], frames_range[1] + 1))
        kind = ("stairs", "paths", "mix", "boxes")[k % 4] if terrain_kind == "parkour" else terrain_kind
        if tile_cells_range is not None:
            dim_x, dim_y = (int(v) for v in rng.integers(tile_cells_range[0], tile_cells_range[1] + 1, size=2))
        else:
            dim_x = dim_y = tile_cells
        size = min(dim_x, dim_y) * dx
        if flat:
            hf = np.zeros((dim_x, dim_y), np.float32)
        elif kind == "boxes":
            hf = box_heightfield(rng, dim_x, dim_y, boxes)
        else:
            assert dim_x == dim_y
            hf = parkour_heightfield(kind, 1000 * seed + k, dim_x, dx)
        min_point = np.array([-dim_x * dx / 2 + dx / 2, -dim_y * dx / 2 + dx / 2], dtype=np.float32)   # cell centres symmetric about 0
        heading = rng.random() * 2 * np.pi
        dur = (nf - 1) / 30.0
        speed = min(1.2, 0.35 * size / max(dur, 1e-3))
        half = 0.5 * speed * dur
        d = np.array([np.cos(heading), np.sin(heading)])
        start = -d * half
        # level a corridor under the path
        ij = ((np.stack(np.meshgrid(np.arange(dim_x), np.arange(dim_y), indexing="ij"), -1)) * dx + min_point)
        rel = ij - start
        along = rel @ d
        perp = np.abs(rel @ np.array([-d[1], d[0]]))
        corridor = (along > -0.8) & (along < 2 * half + 0.8) & (perp < 0.8)
        base = float(np.median(hf)) if not flat else 0.0
        hf[corridor] = base
        fr, con = walking_clip(rng, nf, 30.0, speed, start, heading, base)
        clips.append(dict(frames=fr, contacts=con, fps=30.0, loop=0, weight=1.0, hf=hf, min_point=min_point,
                          dxdy=np.array([dx, dx], np.float32), name="synth_%04d" % k))
    return clips


def tile_square(clips, padding_cells=1):
    """Global heightfield + per-clip xy offsets, laid out like DeepMimicEnv.build_terrain_square
    (envs/ig_parkour/dm_env.py:188-356): ceil(sqrt(M))^2 tiles of equal (max) size, each clip's terrain padded
    with its own minimum height, tile (i, j) filled in clip order along j then i, grid centred on the origin."""
    M = len(clips)
    dx = float(clips[0]["dxdy"][0])
    n_side = int(np.ceil(np.sqrt(M)))
    dim_x = max(c["hf"].shape[0] for c in clips) + 2 * padding_cells
    dim_y = max(c["hf"].shape[1] for c in clips) + 2 * padding_cells
    first_x = -dim_x * n_side * dx / 2.0
    first_y = -dim_y * n_side * dx / 2.0
    hf = np.zeros((dim_x * n_side, dim_y * n_side), dtype=np.float32)
    offsets = np.zeros((M, 1, 2), dtype=np.float32)
    k = 0
    for i in range(n_side):
        for j in range(n_side):
            if k >= M:
                break
            c = clips[k]
            pad = np.pad(c["hf"], padding_cells, constant_values=float(c["hf"].min()))
            local_min = c["min_point"] - dx * padding_cells
            x_off = first_x + i * dim_x * dx
            y_off = first_y + j * dim_y * dx
            offsets[k, 0, 0] = x_off - local_min[0]
            offsets[k, 0, 1] = y_off - local_min[1]
            hf[i * dim_x:i * dim_x + pad.shape[0], j * dim_y:j * dim_y + pad.shape[1]] = pad
            k += 1
    min_point = np.array([first_x, first_y], dtype=np.float32)
    return hf, min_point, np.array([dx, dx], np.float32), offsets
