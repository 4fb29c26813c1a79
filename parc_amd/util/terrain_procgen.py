"""Procedural terrain generators of the tracker datasets (host side, run once per terrain):
boxes, stairs, curvy raised paths and the linear gap / vault parkour course.

Mirrors of the reference's ``util/terrain_util.py`` generators (``add_boxes_to_hf2`` :864-917, ``draw_box`` :971-1000,
``add_stairs_to_hf`` :1002-1043, ``gen_paths_hf`` :544-595, ``linear_parkour_course`` :320-425,
``random_linear_parkour_course`` :427-470; parameters in ``parc_2_kin_gen.py:36-59`` and ``envs/ig_env.py:185-207``).
They draw from the same generators (torch / ``random`` / ``numpy.random``) in the same order with the same fp32
expressions, so a seeded run reproduces the reference's heightfields exactly (tests/golden/g12_procgen.npz).
All of them paint into the heightfield in place.
"""
import random

import numpy as np
import torch


def _rot2(v, angle):
    c, s = torch.cos(angle), torch.sin(angle)
    x, y = v[..., 0], v[..., 1]
    return torch.stack([x * c - y * s, x * s + y * c], dim=-1)


def _inside_rotated_rect(points, center, half_w, half_l, angle):
    """Boolean mask of grid points whose coordinates, rotated by ``angle`` about ``center``, fall strictly inside the
    axis-aligned rectangle ``center +- (half_w, half_l)``."""
    q = _rot2(points - center, angle) + center
    in_x = torch.logical_and(q[..., 0] < center[0] + half_w, q[..., 0] > center[0] - half_w)
    in_y = torch.logical_and(q[..., 1] < center[1] + half_l, q[..., 1] > center[1] - half_l)
    return torch.logical_and(in_x, in_y)


def _grid_points(nx, ny, device, scale=None, origin=None):
    ix = torch.linspace(0, nx - 1, nx, dtype=torch.int64, device=device)
    iy = torch.linspace(0, ny - 1, ny, dtype=torch.int64, device=device)
    if scale is not None:
        ix = ix * scale[0] + origin[0]
        iy = iy * scale[1] + origin[1]
    gx, gy = torch.meshgrid(ix, iy, indexing="ij")
    return torch.stack([gx, gy], dim=-1).to(dtype=torch.float32)


def add_boxes_to_hf2(hf, box_max_height=3.0, box_min_height=-3.0, hf_maxmin=None, num_boxes=32, box_max_len=None, box_min_len=None,
                     max_angle=2.0 * torch.pi, min_angle=0.0):
    """``num_boxes`` randomly placed, sized and rotated boxes in CELL units; each box overwrites the cells it covers with
    one height ~ U[box_min_height, box_max_height]."""
    device = hf.device
    nx, ny = hf.shape
    if box_max_len is None:
        box_max_len = min(nx // 4, ny // 4)
    if box_min_len is None:
        box_min_len = 1
    cells = _grid_points(nx, ny, device)
    extent = torch.tensor(hf.shape, dtype=torch.float32, device=device)
    for _ in range(num_boxes):
        center = torch.rand(size=(2,), dtype=torch.float32, device=device) * extent
        lens = torch.rand(size=(2,), dtype=torch.float32, device=device) * (box_max_len - box_min_len) + box_min_len
        angle = random.random() * (max_angle - min_angle) + min_angle
        inside = _inside_rotated_rect(cells, center, lens[0] / 2, lens[1] / 2, angle * torch.ones_like(cells[..., 0]))
        h = random.random() * (box_max_height - box_min_height) + box_min_height
        hf[...] = torch.where(inside, torch.full_like(hf, h), hf)
    if hf_maxmin is not None:
        hf[...] = torch.clamp(hf, hf_maxmin[..., 1], hf_maxmin[..., 0])


def draw_box(hf, min_point, dxdy, box_center, box_w, box_l, angle, height):
    """One rotated box given in metres (centre, width along x, length along y)."""
    pts = _grid_points(hf.shape[0], hf.shape[1], hf.device, scale=dxdy, origin=min_point)
    hf[_inside_rotated_rect(pts, box_center, box_w / 2, box_l / 2, angle)] = height


def add_stairs_to_hf(terrain, min_stair_start_height=-3.0, max_stair_start_height=1.0, min_step_height=0.15, max_step_height=0.25,
                     num_stairs=1, min_stair_thickness=2.5, max_stair_thickness=8.0):
    """Straight flights of stairs between two random points: one cell-wide step per cell of run, rising by a random
    step height from a random start height."""
    hf = terrain.hf
    lo = terrain.min_point
    span = terrain.get_max_point() - lo
    for _ in range(num_stairs):
        start = torch.rand(size=[2], dtype=torch.float32, device=hf.device) * span + lo
        end = torch.rand(size=[2], dtype=torch.float32, device=hf.device) * span + lo
        run = end - start
        angle = -torch.atan2(run[1], run[0])
        tread = terrain.dxdy[0].item()
        n_steps = int(np.ceil(torch.linalg.norm(run).item() / tread))
        advance = run / n_steps
        h0 = np.random.random() * (max_stair_start_height - min_stair_start_height) + min_stair_start_height
        rise = np.random.random() * (max_step_height - min_step_height) + min_step_height
        thickness = np.random.random() * (max_stair_thickness - min_stair_thickness) + min_stair_thickness
        for j in range(n_steps):
            draw_box(hf, lo, terrain.dxdy, start + j * advance, tread, thickness, angle, h0 + j * rise)


def gen_paths_hf(terrain, num_paths=25, maxpool_size=3, floor_height=-1.0, path_min_height=-0.5, path_max_height=3.0, num_points=1000,
                 curviness=7):
    """Raised curvy paths on a flat floor: every path is a unit-speed random walk in heading (1000 points at 1/30 s) painted at
    one random height, then dilated by a (2*maxpool_size+1)^2 max filter."""
    device = terrain.hf.device
    terrain.hf[...] = floor_height
    dt = 1.0 / 30.0
    hi = terrain.dims * terrain.dxdy + terrain.min_point
    for _ in range(num_paths):
        pos = torch.rand(size=(2,), dtype=torch.float32, device=device) * (hi - terrain.min_point) + terrain.min_point.to(device="cpu")
        vel = torch.randn(size=(2,), dtype=torch.float32, device=device)
        vel[0] = 1.0
        vel = _rot2(vel, torch.rand(size=(1,), dtype=torch.float32, device=device) * 2.0 * torch.pi).squeeze(dim=0)
        track = torch.zeros(size=(num_points, 2), dtype=torch.float32, device=device)
        for i in range(num_points):
            track[i] = pos
            pos = pos + vel * dt
            turn = torch.randn(size=(1,), dtype=torch.float32, device=device)       # one draw per point, as the reference
            vel = _rot2(vel, turn * dt * curviness).squeeze(dim=0)
        ij = torch.round((track - terrain.min_point) / terrain.dxdy).to(dtype=torch.int64)
        ij = torch.clamp(ij, torch.zeros_like(terrain.dims), terrain.dims - 1)
        terrain.hf[ij[:, 0], ij[:, 1]] = random.random() * (path_max_height - path_min_height) + path_min_height
    pool = torch.nn.MaxPool2d(kernel_size=maxpool_size * 2 + 1, stride=1, padding=maxpool_size)
    terrain.hf = pool(terrain.hf.unsqueeze(dim=0)).squeeze(dim=0)


def linear_parkour_course(terrain, block_centers, block_heights, block_dims):
    """Gap / vault course along y: block i sets columns ``[c - w//2, c + w//2]`` of the heightfield to height h for every x.
    Also returns the ribbon mesh the reference feeds the simulator (8 vertices / 8 triangles per block between a front and
    a back edge); the MI355X simulator collides with the heightfield columns directly."""
    n = len(block_centers)
    assert n == len(block_heights) and n == len(block_dims)
    dx, dy = terrain.dxdy[0], terrain.dxdy[1]
    min_x, min_y = terrain.min_point[0], terrain.min_point[1]
    max_x = min_x + terrain.dxdy[0] * terrain.dims[0]
    y_len = terrain.dxdy[1] * terrain.dims[1]
    xl, xr = min_x - dx / 2, max_x + dx / 2
    verts = np.zeros((n * 8 + 4, 3), dtype=np.float32)
    tris = np.zeros((n * 8 + 2, 3), dtype=np.uint32)
    verts[0] = [xl, min_y - dy / 2, 0.0]
    verts[1] = [xr, min_y - dy / 2, 0.0]
    for i in range(n):
        c, h, w = block_centers[i], block_heights[i], block_dims[i]
        j0, j1 = c - w // 2, c + w // 2
        terrain.hf[:, j0:j1 + 1] = h
        front = min_y + j0 * dy - dy / 2
        back = min_y + j1 * dy + dy / 2
        if h == 0.0:        # flat block: keep the four quads non-degenerate
            ys = [front - dy, front, back - dy, back]
            zs = [0.0, 0.0, 0.0, 0.0]
        else:
            ys = [front, front, back, back]
            zs = [0.0, h, h, 0.0]
        v = i * 8 + 2
        for k in range(4):
            verts[v + 2 * k] = [xl, ys[k], zs[k]]
            verts[v + 2 * k + 1] = [xr, ys[k], zs[k]]
        t = i * 8
        for k in range(4):
            a = v + 2 * k
            tris[t + 2 * k] = [a - 2, a - 1, a]
            tris[t + 2 * k + 1] = [a - 1, a + 1, a]
    v = n * 8 + 2
    verts[v] = [xl, y_len + dy / 2, 0.0]
    verts[v + 1] = [xr, y_len + dy / 2, 0.0]
    tris[n * 8] = [v - 2, v - 1, v]
    tris[n * 8 + 1] = [v - 1, v + 1, v]
    return terrain, verts, tris


def random_linear_parkour_course(terrain, gap_width, gap_height, vault_width, vault_height, num_padding_cells):
    """Blocks every 6.5 m or 8 m (coin flip) along y until the terrain ends, each a vault or a gap (coin flip)."""
    centers = []
    y, y_end, dy = 0.0, terrain.get_real_size()[1], terrain.dxdy[1]
    while y < y_end:
        y += 6.5 if random.random() < 0.5 else 8.0
        centers.append(int(round(y / dy)) + num_padding_cells)
    centers = np.array(centers, dtype=np.int64)
    kinds = np.random.randint(0, 2, size=(centers.shape[0],))
    heights = np.array([vault_height, gap_height])[kinds]
    widths = np.array([vault_width, gap_width])[kinds]
    return linear_parkour_course(terrain, centers, heights, widths)
