"""Heightfield container of the tracker.

Mirror of the hot-path part of the reference's ``util/terrain_util.py``: ``SubTerrain`` (:21-258) with the
same field names -- motion pickles embed instances of it, so it must unpickle under the module path
``util.terrain_util`` (parc_amd.install_reference_aliases) -- and ``get_local_hf_from_terrain``
(:1329-1346).  Procedural generators and the voxel-mesh export are "next" rows.
"""
import contextlib
import copy
import pickle
import sys

import numpy as np
import torch

from . import torch_util


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

    def flip_by_XZ_axis(self):
        """mirror the field in the plane y = 0 (reference util/terrain_util.py:169-178): cells reversed along y, the new first
        column centre is minus the old last one"""
        max_point = self.get_max_point()
        self.hf = torch.flip(self.hf, dims=[1])
        self.hf_mask = torch.flip(self.hf_mask, dims=[1])
        self.hf_maxmin = torch.flip(self.hf_maxmin, dims=[1])
        self.min_point[1] = -1.0 * max_point[1]

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


# Motion files and the terrain.pkl cache embed SubTerrain instances, and the reference reads them with a plain
# pickle.load (anim/motion_lib.py:240, envs/ig_parkour/dm_env.py:136,161,496), which resolves the class by the module path
# stored in the file.  The class therefore names the reference's path.  Nothing is registered in sys.modules at import time (a process
# that imports this package next to the reference, or next to another project's top-level `util`, must not have its imports
# redirected): the path resolves to this module inside `reference_pickle_path()` / `dump_reference_pickle()` - the writers of this
# package - or after `parc_amd.install_reference_aliases()`; a plain `pickle.dump` of a SubTerrain anywhere else raises PicklingError.
REFERENCE_MODULE = "util.terrain_util"
SubTerrain.__module__ = REFERENCE_MODULE


# other modules whose classes are written into motion files under a reference path: reference module name -> module
# (tools.motion_opt.motion_optimization registers BodyConstraint here)
_REFERENCE_PICKLE_MODULES = {}


def register_reference_pickle_module(reference_name, module):
    _REFERENCE_PICKLE_MODULES[reference_name] = module


@contextlib.contextmanager
def reference_pickle_path():
    """Inside this context ``pickle.dump`` of a SubTerrain always succeeds and writes ``util.terrain_util SubTerrain``,
    even in a process where another ``util.terrain_util`` (e.g. the reference's own) is already imported."""
    names = ["util", REFERENCE_MODULE]
    for ref_name in _REFERENCE_PICKLE_MODULES:
        parts = ref_name.split(".")
        names += [".".join(parts[:k]) for k in range(1, len(parts) + 1)]
    prev = {k: sys.modules.get(k) for k in names}
    sys.modules[REFERENCE_MODULE] = sys.modules[__name__]
    if prev["util"] is None:
        sys.modules["util"] = sys.modules[__name__.rpartition(".")[0]]
    for ref_name, mod in _REFERENCE_PICKLE_MODULES.items():
        parts = ref_name.split(".")
        for k in range(1, len(parts)):
            pkg = ".".join(parts[:k])
            if sys.modules.get(pkg) is None:
                import types
                ns = types.ModuleType(pkg)
                ns.__path__ = []
                sys.modules[pkg] = ns
        sys.modules[ref_name] = mod
    try:
        yield
    finally:
        for k, v in prev.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def dump_reference_pickle(obj, path):
    """Write ``obj`` (a motion dict / terrain cache holding SubTerrain instances) the way the reference does:
    ``pickle.dump`` with the default protocol, classes under the reference's module names."""
    with reference_pickle_path(), open(path, "wb") as f:
        pickle.dump(obj, f)


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


# ---------------------------------------------------------------------------------------------------------------------
# Terrain / body-geometry queries around the tracker ("next" rows): penetration distance of point
# sets into a heightfield (HIP kernel parc_points_hf_sdf) and the per-clip heightfield preprocessing of the dataset
# builder.  The character's pose comes from the same FK kernels the tracker uses (KinCharModel.dof_to_rot /
# forward_kinematics); `points_hf_sdf` / `motion_frames_hf_sdf_loss` are differentiable (kernel for the arg-min column, torch ops for the selected
# branch and for the pose), the preprocessing functions are forward-only.
# ---------------------------------------------------------------------------------------------------------------------
class HfGrid:
    """What points_hf_sdf derives from a heightfield batch before it can launch: cell-centre coordinates relative to cell (0, 0) (torch's
    own linspace, evaluated on the host: X + Y values) and the half cell size.  Callers that query the same heightfields many times
    (the motion optimiser: thousands of iterations, captured in a hipGraph) prepare it once and pass it as ``grid=``."""

    def __init__(self, hf, hf_dxdy, device=None):
        X, Y = int(hf.shape[-2]), int(hf.shape[-1])
        dev = hf.device if device is None else device
        dxdy = hf_dxdy.detach().to(torch.float32).cpu()
        self.xs = torch.linspace(0.0, (X - 1.0) * dxdy[0].item(), X).to(dev)
        self.ys = torch.linspace(0.0, (Y - 1.0) * dxdy[1].item(), Y).to(dev)
        half = dxdy / 2.0
        self.half = (float(half[0]), float(half[1]))
        self.shape = (X, Y)


def points_hf_sdf(points, hf, hf_min_box_center, hf_dxdy, base_z=-10.0, inverted=True, radius=None, grid=None):
    """Signed distance of points [B, N, 3] to heightfields hf [B, X, Y] made of dx x dy columns whose cell (0, 0) is centred at
    hf_min_box_center [B, 2]; inverted (default) = negative depth below the surface for points in the ground.
    Reference: util/terrain_util.py:1835-1893.  One launch, no [B, N, X*Y, 3] temporaries.

    Differentiable in ``points``: the kernel also reports WHICH column attains the minimum, and the adjoint is the derivative of the
    distance to that one column - the branch torch.min would have routed the gradient through - in one launch (parc_points_hf_sdf_grad).
    When the heightfield itself requires grad, that column's distance is re-evaluated with torch ops instead.  With ``grid=HfGrid(...)``
    the call reads nothing back from the device."""
    assert points.dim() == 3 and hf.dim() == 3 and hf_min_box_center.dim() == 2
    B, N = int(points.shape[0]), int(points.shape[1])
    assert hf.shape[0] == B and hf_min_box_center.shape[0] == B
    X, Y = int(hf.shape[1]), int(hf.shape[2])
    dev = points.device
    if grid is None:
        assert hf_dxdy.dim() == 1
        grid = HfGrid(hf, hf_dxdy, dev)
    assert grid.shape == (X, Y)
    xs, ys, half = grid.xs, grid.ys, grid.half
    if radius is not None:
        assert isinstance(radius, float) and radius > 0.0
    want_grad = torch.is_grad_enabled() and (points.requires_grad or hf.requires_grad)
    if want_grad and not (hf.requires_grad or hf_min_box_center.requires_grad):
        # the common case (poses are optimised, the terrain is fixed): the adjoint is one launch too
        return _PointsHfSdf.apply(points, hf, hf_min_box_center, grid, float(base_z), bool(inverted), radius)
    out, cell = _points_hf_sdf_launch(points, hf, hf_min_box_center, grid, base_z, inverted, radius, want_grad)
    if not want_grad:
        return out
    # gradients with respect to the heightfield asked for: the selected column per point, through autograd (same expressions as the
    # kernel / the reference)
    ci = cell.long()
    i, j = torch.div(ci, Y, rounding_mode="floor"), ci % Y
    h = torch.gather(hf.reshape(B, -1), 1, ci)
    cx = xs[i] + hf_min_box_center[:, 0:1]
    cy = ys[j] + hf_min_box_center[:, 1:2]
    if inverted:
        top = -base_z
        cz, hz = (h + top) / 2.0, (top - h) / 2.0
    else:
        cz, hz = (h + base_z) / 2.0, (h - base_z) / 2.0
    q = torch.stack([(points[..., 0] - cx).abs() - float(half[0]), (points[..., 1] - cy).abs() - float(half[1]), (points[..., 2] - cz).abs() - hz],
                    dim=-1)
    sd = torch.linalg.vector_norm(q.clamp(min=0.0), dim=-1) + q.max(dim=-1)[0].clamp(max=0.0)
    if radius is not None:
        sd = sd - radius
    return -sd if inverted else sd


def _points_hf_sdf_launch(points, hf, hf_min_box_center, grid, base_z, inverted, radius, want_cell):
    from .. import _hip
    B, N = int(points.shape[0]), int(points.shape[1])
    X, Y = grid.shape
    dev = points.device
    out = torch.empty((B, N), dtype=torch.float32, device=dev)
    cell = torch.empty((B, N), dtype=torch.int32, device=dev) if want_cell else None
    pts = points.detach().to(torch.float32).contiguous()
    hfc = hf.detach().to(torch.float32).contiguous()
    mbc = hf_min_box_center.detach().to(torch.float32).contiguous()
    _hip.check(_hip.lib().parc_points_hf_sdf(_hip.stream(), B, N, X, Y, _hip.ptr(pts), _hip.ptr(hfc), _hip.ptr(mbc), _hip.ptr(grid.xs), _hip.ptr(grid.ys),
                                             float(grid.half[0]), float(grid.half[1]), float(base_z), 1 if inverted else 0,
                                             float(radius) if radius is not None else 0.0, _hip.ptr(out), _hip.ptr(cell)), "parc_points_hf_sdf")
    return out, cell


class _PointsHfSdf(torch.autograd.Function):
    """points_hf_sdf for a fixed terrain: forward = the query (which also reports the arg-min column), backward = parc_points_hf_sdf_grad
    (g_points = g_out * d(distance to that column)/d(point)), one launch each."""

    @staticmethod
    def forward(ctx, points, hf, hf_min_box_center, grid, base_z, inverted, radius):
        out, cell = _points_hf_sdf_launch(points, hf, hf_min_box_center, grid, base_z, inverted, radius, True)
        ctx.save_for_backward(points.detach().to(torch.float32).contiguous(), hf.detach().to(torch.float32).contiguous(),
                              hf_min_box_center.detach().to(torch.float32).contiguous(), cell)
        ctx.grid, ctx.base_z, ctx.inverted = grid, base_z, inverted
        return out

    @staticmethod
    def backward(ctx, g_out):
        from .. import _hip
        pts, hfc, mbc, cell = ctx.saved_tensors
        grid = ctx.grid
        B, N = int(pts.shape[0]), int(pts.shape[1])
        X, Y = grid.shape
        g = g_out.to(torch.float32).contiguous()
        g_pts = torch.empty_like(pts)
        _hip.check(_hip.lib().parc_points_hf_sdf_grad(_hip.stream(), B, N, X, Y, _hip.ptr(pts), _hip.ptr(hfc), _hip.ptr(mbc), _hip.ptr(grid.xs),
                                                      _hip.ptr(grid.ys), float(grid.half[0]), float(grid.half[1]), float(ctx.base_z),
                                                      1 if ctx.inverted else 0, _hip.ptr(cell), _hip.ptr(g), _hip.ptr(g_pts)),
                   "parc_points_hf_sdf_grad")
        return g_pts, None, None, None, None, None, None


class BodyPoints:
    """The sample points of all bodies as one table: local coordinates [P, 3], owning body [P] and the offsets of each body's
    (contiguous) range - prepared once per character / point set."""

    def __init__(self, char_point_samples, device):
        counts = [int(p.shape[0]) for p in char_point_samples]
        self.num_bodies = len(counts)
        self.counts = counts
        self.start = [sum(counts[:b]) for b in range(len(counts) + 1)]
        self.local = torch.cat([p.to(device=device, dtype=torch.float32).reshape(-1, 3) for p in char_point_samples], dim=0).contiguous()
        self.owner = torch.cat([torch.full((n,), b, dtype=torch.int64, device=device) for b, n in enumerate(counts)])
        self.owner32 = self.owner.to(torch.int32).contiguous()
        self.start32 = torch.tensor(self.start, dtype=torch.int32, device=device)
        self.num_points = int(self.local.shape[0])

    def world(self, body_pos, body_rot):
        """body_pos [..., B, 3], body_rot [..., B, 4] -> world positions [..., P, 3] of every sample point (differentiable: forward and
        adjoint are one launch each, parc_body_points_world / _grad)"""
        lead = body_pos.shape[:-2]
        w = _BodyPointsWorld.apply(body_pos.reshape(-1, self.num_bodies, 3), body_rot.reshape(-1, self.num_bodies, 4), self)
        return w.reshape(lead + (self.num_points, 3))


class _BodyPointsWorld(torch.autograd.Function):
    @staticmethod
    def forward(ctx, body_pos, body_rot, bp):
        from .. import _hip
        T_ = int(body_pos.shape[0])
        pos, rot = body_pos.detach().to(torch.float32).contiguous(), body_rot.detach().to(torch.float32).contiguous()
        world = torch.empty((T_, bp.num_points, 3), dtype=torch.float32, device=pos.device)
        _hip.check(_hip.lib().parc_body_points_world(_hip.stream(), T_, bp.num_bodies, bp.num_points, _hip.ptr(pos), _hip.ptr(rot), _hip.ptr(bp.local),
                                                     _hip.ptr(bp.owner32), _hip.ptr(world)), "parc_body_points_world")
        ctx.save_for_backward(rot)
        ctx.bp = bp
        return world

    @staticmethod
    def backward(ctx, g_world):
        from .. import _hip
        rot, = ctx.saved_tensors
        bp = ctx.bp
        T_ = int(rot.shape[0])
        g = g_world.to(torch.float32).contiguous()
        g_pos = torch.empty((T_, bp.num_bodies, 3), dtype=torch.float32, device=rot.device)
        g_rot = torch.empty((T_, bp.num_bodies, 4), dtype=torch.float32, device=rot.device)
        _hip.check(_hip.lib().parc_body_points_world_grad(_hip.stream(), T_, bp.num_bodies, bp.num_points, _hip.ptr(rot), _hip.ptr(bp.local),
                                                          _hip.ptr(bp.start32), _hip.ptr(g), _hip.ptr(g_pos), _hip.ptr(g_rot)),
                   "parc_body_points_world_grad")
        return g_pos, g_rot, None


def _body_points_world(motion_frames, char_model, char_point_samples):
    """frames [..., 34] -> (world positions [..., P, 3] of every sample point, owning body [P]); bodies in order, a body's
    points in the order of its sample tensor."""
    bp = BodyPoints(char_point_samples, motion_frames.device)
    if torch.is_grad_enabled() and motion_frames.requires_grad:
        # the pose chain with its one-launch adjoint, so that autograd reaches the frames (KinCharModel.pose_chain)
        lead = motion_frames.shape[:-1]
        flat = motion_frames.reshape(-1, motion_frames.shape[-1])
        _, _, body_pos, body_rot = char_model.pose_chain(flat[:, 0:3], flat[:, 3:6], flat[:, 6:])
        body_pos, body_rot = body_pos.reshape(lead + body_pos.shape[1:]), body_rot.reshape(lead + body_rot.shape[1:])
    else:
        root_rot = torch_util.exp_map_to_quat(motion_frames[..., 3:6])
        joint_rot = char_model.dof_to_rot(motion_frames[..., 6:])
        body_pos, body_rot = char_model.forward_kinematics(motion_frames[..., 0:3], root_rot, joint_rot)
    return bp.world(body_pos, body_rot), bp.owner


def motion_frames_hf_sdf_loss(motion_frames, char_point_samples, hf, hf_min_box_center, hf_dxdy, char_model, ret_vis_info=False,
                              interior_distance=True):
    """0.5 * sum over all body sample points and frames of the squared terrain penetration of motion_frames [B, T, 34] (frames in
    the heightfields' coordinate frame).  Reference: util/terrain_util.py:1895-1951; point order of the returned arrays as there
    (body-major: all frames of body 0's points, then body 1's, ...)."""
    B, T = motion_frames.shape[0], motion_frames.shape[1]
    world, owner = _body_points_world(motion_frames, char_model, char_point_samples)          # [B, T, P, 3]
    blocks, start = [], 0
    for p in char_point_samples:
        n = p.shape[0]
        blocks.append(world[:, :, start:start + n].reshape(B, T * n, 3))
        start += n
    pts = torch.cat(blocks, dim=1)
    sdf = points_hf_sdf(pts, hf, hf_min_box_center, hf_dxdy, base_z=-10.0, inverted=interior_distance)
    pen = sdf.clamp(max=0.0) if interior_distance else sdf.clamp(min=0.0)
    loss = 0.5 * (pen * pen).sum(dim=-1)
    return (loss, pts, sdf) if ret_vis_info else loss


def compute_hf_mask_inds(motion_frames, terrain, char_model, char_body_points):
    """For a clip [T, 34] on its terrain: per frame the (sorted, unique) grid cells under any body sample point, and per cell
    the lowest sample-point height over the whole clip (99999.9999 where nothing passes).  Clears terrain.hf_mask like the
    reference (util/terrain_util.py:1953-2000), whose per-frame / per-body / per-point Python loops become one FK launch, one
    scatter-min and one sort."""
    T = motion_frames.shape[0]
    world, _ = _body_points_world(motion_frames, char_model, char_body_points)                 # [T, P, 3]
    terrain.hf_mask[...] = False
    g = terrain.get_grid_index(world[..., 0:2])                                                 # [T, P, 2]
    Y = terrain.hf.shape[1]
    flat = g[..., 0] * Y + g[..., 1]
    lowest = torch.full((terrain.hf.numel(),), 99999.9999, dtype=terrain.hf.dtype, device=terrain.hf.device)
    lowest.scatter_reduce_(0, flat.reshape(-1), world[..., 2].reshape(-1).to(lowest.dtype), reduce="amin", include_self=True)
    # unique cells per frame: sort the (frame, cell) keys once, drop repeats, split by frame
    keys = torch.unique(torch.arange(T, device=flat.device).unsqueeze(1) * terrain.hf.numel() + flat)
    frame_of = torch.div(keys, terrain.hf.numel(), rounding_mode="floor")
    cell = keys - frame_of * terrain.hf.numel()
    ij = torch.stack([torch.div(cell, Y, rounding_mode="floor"), cell % Y], dim=-1)
    counts = torch.bincount(frame_of, minlength=T).tolist()
    return list(torch.split(ij, counts)), lowest.reshape(terrain.hf.shape)


def compute_hf_mask_from_inds(terrain, mask_grid_inds):
    mask = torch.zeros_like(terrain.hf_mask)
    if len(mask_grid_inds) > 0:
        ij = torch.cat(list(mask_grid_inds), dim=0)
        mask[ij[:, 0], ij[:, 1]] = True
    return mask


def compute_hf_mask(motion_frames, terrain, char_model, char_body_points):
    inds, _ = compute_hf_mask_inds(motion_frames, terrain, char_model, char_body_points)
    return compute_hf_mask_from_inds(terrain, inds)


def compute_hf_extra_vals(motion_frames, terrain, char_model, char_body_points, z_buf=3.0, jump_buf=0.8):
    """Dataset preprocessing of one clip (util/terrain_util.py:2017-2052): hf_mask = cells the character passes over;
    hf_maxmin = allowed height band per cell - pinned to the terrain under the character, [lowest terrain - z_buf, highest root +
    z_buf] elsewhere, and capped at (lowest body point - jump_buf) where the character flies at least jump_buf above the ground.
    Returns the per-frame cell lists; terrain is updated in place."""
    inds, lowest = compute_hf_mask_inds(motion_frames, terrain, char_model, char_body_points)
    terrain.hf_mask = compute_hf_mask_from_inds(terrain, inds)
    top = torch.max(motion_frames[:, 2]).item() + z_buf
    bottom = torch.min(terrain.hf).item() - z_buf
    under = terrain.hf_mask
    airborne = torch.logical_and(lowest - terrain.hf >= jump_buf, under)
    hi = torch.where(under, terrain.hf, torch.full_like(terrain.hf, top))
    lo = torch.where(under, terrain.hf, torch.full_like(terrain.hf, bottom))
    terrain.hf_maxmin[..., 0] = torch.where(airborne, lowest - jump_buf, hi)
    terrain.hf_maxmin[..., 1] = torch.where(airborne, torch.full_like(terrain.hf, bottom), lo)
    return inds
