"""Heightfield container of the tracker.

Mirror of the hot-path part of the reference's ``util/terrain_util.py``: ``SubTerrain`` (:21-258) with the
same field names -- motion pickles embed instances of it, so it must unpickle under the module path
``util.terrain_util`` (parc_amd.install_reference_aliases) -- and ``get_local_hf_from_terrain``
(:1329-1346).  Procedural generators and the voxel-mesh export are "next" rows.
"""
import copy

import numpy as np
import torch


class SubTerrain:
    def __init__(self, terrain_name="terrain", x_dim=256, y_dim=256, dx=1.0, dy=1.0, min_x=-1.0, min_y=-1.0, device="cuda:0"):
        self.terrain_name = terrain_name
        self.hf = torch.zeros((x_dim, y_dim), dtype=torch.float32, device=device)
        self.dims = torch.tensor([x_dim, y_dim], dtype=torch.int64, device=device)
        self.min_point = torch.tensor([min_x, min_y], dtype=torch.float32, device=device)
        self.dxdy = torch.tensor([dx, dy], dtype=torch.float32, device=device)
        self.hf_mask = torch.zeros((x_dim, y_dim), dtype=torch.bool, device=device)
        self.hf_maxmin = torch.zeros((x_dim, y_dim, 2), dtype=torch.float32, device=device)
        self.hf_maxmin[..., 0] = 1.0
        self.hf_maxmin[..., 1] = -1.0

    _FIELDS = (("hf", torch.float32), ("dims", torch.int64), ("min_point", torch.float32), ("dxdy", torch.float32),
               ("hf_mask", torch.bool), ("hf_maxmin", torch.float32))

    @classmethod
    def from_arrays(cls, hf, min_point, dxdy, hf_mask=None, hf_maxmin=None, name="terrain", device="cpu"):
        hf = np.asarray(hf, dtype=np.float32)
        t = cls(name, hf.shape[0], hf.shape[1], float(dxdy[0]), float(dxdy[1]), float(min_point[0]), float(min_point[1]), device=device)
        t.hf[:] = torch.as_tensor(hf, device=device)
        if hf_mask is not None:
            t.hf_mask[:] = torch.as_tensor(np.asarray(hf_mask, dtype=bool), device=device)
        if hf_maxmin is not None:
            t.hf_maxmin[:] = torch.as_tensor(np.asarray(hf_maxmin, dtype=np.float32), device=device)
        return t

    def update_old(self):
        if not hasattr(self, "hf_maxmin"):
            shape = (self.hf.shape[0], self.hf.shape[1], 2)
            if isinstance(self.hf, torch.Tensor):
                self.hf_maxmin = torch.zeros(shape, dtype=torch.float32, device=self.hf.device)
            else:
                self.hf_maxmin = np.zeros(shape, dtype=np.float32)
            self.hf_maxmin[..., 0] = 1.0
            self.hf_maxmin[..., 1] = -1.0

    def to_torch(self, device):
        for name, dtype in self._FIELDS:
            v = getattr(self, name)
            if isinstance(v, torch.Tensor):
                setattr(self, name, v.to(device=device))
            else:
                setattr(self, name, torch.as_tensor(np.asarray(v), dtype=dtype, device=device))

    def set_device(self, device):
        self.to_torch(device)

    def to_numpy(self):
        for name, _ in self._FIELDS:
            v = getattr(self, name)
            if isinstance(v, torch.Tensor):
                setattr(self, name, v.detach().cpu().numpy())

    def numpy_copy(self):
        t = copy.deepcopy(self)
        t.to_numpy()
        return t

    def torch_copy(self):
        return copy.deepcopy(self)

    def get_real_size(self):
        return self.dims * self.dxdy

    def get_max_point(self):
        return self.min_point + self.get_real_size() - self.dxdy

    def get_inbounds_grid_index(self, grid_ind):
        return torch.clamp(grid_ind, torch.zeros_like(self.dims), self.dims - 1)

    def round_point_to_grid_index(self, point):
        return torch.round((point - self.min_point) / self.dxdy).to(dtype=torch.int64)

    def get_grid_index(self, point):
        return self.get_inbounds_grid_index(self.round_point_to_grid_index(point))

    def get_hf_val_from_points(self, xy_points):
        g = self.get_grid_index(xy_points)
        return self.hf[g[..., 0], g[..., 1]]

    def get_point(self, ij):
        return self.min_point + ij * self.dxdy

    def pad(self, padding_size, height=0.0):
        p = padding_size
        self.hf = torch.nn.functional.pad(self.hf, [p, p, p, p], value=height)
        mask = torch.zeros((self.hf_mask.shape[0] + 2 * p, self.hf_mask.shape[1] + 2 * p), dtype=torch.bool, device=self.hf.device)
        mask[p:mask.shape[0] - p, p:mask.shape[1] - p] = self.hf_mask
        self.hf_mask = mask
        mx = torch.max(self.hf_maxmin[..., 0]).item()
        mn = torch.min(self.hf_maxmin[..., 1]).item()
        new_max = torch.nn.functional.pad(self.hf_maxmin[..., 0], [p, p, p, p], value=mx)
        new_min = torch.nn.functional.pad(self.hf_maxmin[..., 1], [p, p, p, p], value=mn)
        self.hf_maxmin = torch.stack([new_max, new_min], dim=-1)
        self.min_point = self.min_point - self.dxdy * p
        self.dims = self.dims + 2 * p


def get_local_hf_from_terrain(xy_points, terrain):
    """Nearest-cell height lookup for arbitrary query points (torch; the per-step 441-point fan goes through
    the HIP kernel parc_refresh_obs_hfs instead)."""
    g = terrain.get_grid_index(xy_points)
    return terrain.hf[g[..., 0], g[..., 1]]


def slice_terrain_around_motion(motion_frames, terrain, padding=1.0):
    """Cut the heightfield window under a recorded motion and re-centre both on the first frame
    (reference: util/terrain_util.py:1675-1759, localize=True).  motion_frames: numpy [T, >=3] with global xy;
    returns (SubTerrain on CPU, localized numpy frames)."""
    frames = np.array(motion_frames, dtype=np.float32, copy=True)
    t = terrain.torch_copy()
    t.set_device("cpu")
    mn = torch.tensor([frames[:, 0].min(), frames[:, 1].min()], dtype=torch.float32) - padding
    mx = torch.tensor([frames[:, 0].max(), frames[:, 1].max()], dtype=torch.float32) + padding
    gmin = torch.round((mn - t.min_point) / t.dxdy) * t.dxdy + t.min_point
    gmax = torch.round((mx - t.min_point) / t.dxdy) * t.dxdy + t.min_point
    xs = torch.arange(gmin[0].item(), (gmax[0] + t.dxdy[0]).item(), step=t.dxdy[0].item())
    ys = torch.arange(gmin[1].item(), (gmax[1] + t.dxdy[1]).item(), step=t.dxdy[1].item())
    x, y = torch.meshgrid(xs, ys, indexing="ij")
    g = t.get_grid_index(torch.stack([x, y], dim=-1))
    canon = frames[0, 0:2].copy()
    frames[:, 0:2] -= canon
    out = SubTerrain("terrain", g.shape[0], g.shape[1], t.dxdy[0].item(), t.dxdy[1].item(), gmin[0].item() - float(canon[0]),
                     gmin[1].item() - float(canon[1]), device="cpu")
    out.hf = t.hf[g[..., 0], g[..., 1]].clone()
    out.hf_mask = t.hf_mask[g[..., 0], g[..., 1]].clone()
    out.hf_maxmin = t.hf_maxmin[g[..., 0], g[..., 1]].clone()
    z0 = out.get_hf_val_from_points(torch.tensor(frames[0, 0:2])).item()
    frames[:, 2] -= z0
    out.hf = out.hf - z0
    return out, frames


def convert_heightfield_to_voxelized_trimesh(hf, min_x, min_y, dx=0.1, padding=None):
    """Heightfield -> the flat-topped-column triangle mesh the reference hands to the simulator
    (util/terrain_util.py:1099-1251): 4 vertices + 2 triangles per cell top, 2 triangles per shared cell edge along x and
    along y (degenerate where neighbours are level), optionally an 8-vertex / 8-triangle skirt ``padding`` metres wide
    at the lowest height.  Same vertex / triangle order and the same float64 -> float32 arithmetic as the reference's
    per-cell Python loops, as whole-array numpy (O(cells) array ops instead of O(cells) interpreter iterations).
    Returns (vertices float32 [V,3], tris uint32 [T,3])."""
    if isinstance(hf, torch.Tensor):
        hf = hf.detach().cpu().numpy()
    hf = np.asarray(hf)
    nx, ny = hf.shape
    pad = padding is not None and padding > 0.0
    n_cells = nx * ny
    n_flat, n_x, n_y = 2 * n_cells, 2 * (nx - 1) * ny, 2 * nx * (ny - 1)
    vertices = np.zeros((4 * n_cells + (8 if pad else 0), 3), dtype=np.float32)
    tris = np.zeros((n_flat + n_x + n_y + (8 if pad else 0), 3), dtype=np.uint32)
    x = dx * np.arange(nx, dtype=np.float64) + min_x           # cell centres, float64 like the reference's Python floats
    y = dx * np.arange(ny, dtype=np.float64) + min_y
    xm, xp, ym, yp = x - dx / 2, x + dx / 2, y - dx / 2, y + dx / 2
    V = vertices[:4 * n_cells].reshape(nx, ny, 4, 3)
    V[:, :, 0, 0] = xm[:, None]; V[:, :, 0, 1] = ym[None, :]
    V[:, :, 1, 0] = xm[:, None]; V[:, :, 1, 1] = yp[None, :]
    V[:, :, 2, 0] = xp[:, None]; V[:, :, 2, 1] = yp[None, :]
    V[:, :, 3, 0] = xp[:, None]; V[:, :, 3, 1] = ym[None, :]
    V[:, :, :, 2] = hf[:, :, None]
    cell = (np.arange(nx, dtype=np.int64)[:, None] * ny + np.arange(ny, dtype=np.int64)[None, :])
    c4 = (cell * 4).reshape(-1)
    F = tris[:n_flat].reshape(n_cells, 2, 3)
    F[:, 0] = np.stack([c4 + 0, c4 + 2, c4 + 1], axis=-1)
    F[:, 1] = np.stack([c4 + 0, c4 + 3, c4 + 2], axis=-1)
    if nx > 1:
        a = (cell[:-1] * 4).reshape(-1)                          # cell (i, j) and its +x neighbour (i+1, j)
        b = (cell[1:] * 4).reshape(-1)
        v1, v2, v3, v4 = a + 3, a + 2, b + 1, b + 0
        X = tris[n_flat:n_flat + n_x].reshape(-1, 2, 3)
        X[:, 0] = np.stack([v1, v3, v2], axis=-1)
        X[:, 1] = np.stack([v1, v4, v3], axis=-1)
    if ny > 1:
        a = (cell[:, :-1] * 4).reshape(-1)                       # cell (i, j) and its +y neighbour (i, j+1)
        b = (cell[:, 1:] * 4).reshape(-1)
        v1, v2, v3, v4 = a + 1, a + 2, b + 3, b + 0
        Y = tris[n_flat + n_x:n_flat + n_x + n_y].reshape(-1, 2, 3)
        Y[:, 0] = np.stack([v1, v2, v3], axis=-1)
        Y[:, 1] = np.stack([v1, v3, v4], axis=-1)
    if pad:
        max_x = min_x + dx * (nx - 1)
        max_y = min_y + dx * (ny - 1)
        z = np.min(hf)
        p0 = np.array([min_x - dx / 2, min_y - dx / 2, z])
        p1 = np.array([max_x + dx / 2, min_y - dx / 2, z])
        p2 = np.array([min_x - dx / 2, max_y + dx / 2, z])
        p3 = np.array([max_x + dx / 2, max_y + dx / 2, z])
        vertices[-8], vertices[-7], vertices[-6], vertices[-5] = p0, p1, p2, p3
        vertices[-4] = p0 + np.array([-padding, -padding, 0.0])
        vertices[-3] = p1 + np.array([+padding, -padding, 0.0])
        vertices[-2] = p2 + np.array([-padding, +padding, 0.0])
        vertices[-1] = p3 + np.array([+padding, +padding, 0.0])
        n = vertices.shape[0]
        v0, v1, v2, v3, v4, v5, v6, v7 = (n - 8 + k for k in range(8))
        tris[-8:] = [[v0, v4, v5], [v0, v5, v1], [v1, v5, v7], [v1, v7, v3], [v3, v7, v6], [v3, v6, v2], [v2, v6, v4], [v2, v4, v0]]
    return vertices, tris


# procedural generators under the reference's names (util/terrain_util.py)
from .terrain_procgen import (add_boxes_to_hf2, add_stairs_to_hf, draw_box, gen_paths_hf, linear_parkour_course,  # noqa: E402,F401
                              random_linear_parkour_course)
