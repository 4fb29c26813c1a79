"""Ray-fan template of the local heightmap observation (reference: util/geom_util.py:249-270)."""
import math

import numpy as np
import torch


def get_xy_points_cone(center, dx, num_neg, num_pos, num_rays_neg, num_rays_pos, angle_between_rays):
    """[num_rays * (num_neg + num_pos + 1), 2] template points: `num_rays` straight rays through the origin,
    ray r rotated by -angle*(num_rays_neg - r).  Computed once on the host in fp32 with torch's linspace."""
    device = center.device if isinstance(center, torch.Tensor) else "cpu"
    dim = num_neg + num_pos + 1
    x = torch.linspace(-dx * num_neg, dx * num_pos, dim, dtype=torch.float32)
    num_rays = num_rays_neg + 1 + num_rays_pos
    rays = []
    for r in range(num_rays):
        ang = torch.tensor(-angle_between_rays * (num_rays_neg - r), dtype=torch.float32)
        c, s = torch.cos(ang), torch.sin(ang)
        rays.append(torch.stack([x * c - 0.0 * s, x * s + 0.0 * c], dim=-1))
    return torch.cat(rays, dim=0).to(device)


# ---------------------------------------------------------------------------------------------------------------------
# Surface point samples of the character's collision primitives (body frame), the point sets the terrain-penetration loss
# and the heightfield-mask preprocessing push through FK.  Behavioural contract: the reference's
# util/geom_util.py:725-869 (get_box_point_surface_samples, get_sphere_point_surface_samples,
# get_capsule_point_surface_samples, get_char_point_samples) - same point order and the same fp32 operation order, so the
# sets agree to the bit for boxes and capsules (fixture g13).  Sphere points are the 12 vertices of a unit icosahedron scaled
# by the radius, which is what trimesh.creation.icosphere(subdivisions=0) yields in the reference; trimesh is not available
# here, so that vertex ORDER is restated from its published construction and is "parity unpinned".
# ---------------------------------------------------------------------------------------------------------------------
def _unit(n, device):
    return torch.linspace(0.0, 1.0, n, device=device)


def get_box_point_surface_samples(box_halfdims, device, num_slices=2, dim_x=6, dim_y=3):
    """dim_x x dim_y lattice on each of num_slices horizontal cuts from -hz to +hz; order: slice, then x, then y."""
    h = box_halfdims
    x = _unit(dim_x, device) * h[0] * 2.0 - h[0]
    y = _unit(dim_y, device) * h[1] * 2.0 - h[1]
    z = _unit(num_slices, device) * h[2] * 2.0 - h[2]
    pts = torch.empty((num_slices, dim_x, dim_y, 3), dtype=torch.float32, device=device)
    pts[..., 0] = x[None, :, None]
    pts[..., 1] = y[None, None, :]
    pts[..., 2] = z[:, None, None]
    return pts.reshape(-1, 3)


_ICOSAHEDRON = None


def _icosahedron_vertices():
    """The 12 vertices (0, +-1, +-phi) and cyclic permutations, in the order trimesh's icosahedron() lists them,
    normalised to the unit sphere (float64)."""
    global _ICOSAHEDRON
    if _ICOSAHEDRON is None:
        t = (1.0 + 5.0 ** 0.5) / 2.0
        v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                      [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
        _ICOSAHEDRON = v / np.linalg.norm(v, axis=1, keepdims=True)
    return _ICOSAHEDRON


def get_sphere_point_surface_samples(radius, device, num_subdivisions=0):
    if num_subdivisions != 0:
        raise NotImplementedError("only the un-subdivided icosphere (12 points) is used by the character samplers")
    return torch.from_numpy(_icosahedron_vertices() * float(radius)).to(device=device, dtype=torch.float32)


def get_capsule_point_surface_samples(capsule_length, capsule_radius, device, num_cylinder_slices=3, num_circle_points=4,
                                      num_sphere_subdivisons=0, ignore_hemispheres=True):
    """Rings of num_circle_points on num_cylinder_slices cuts of the cylinder part (axis = z, centred); order: ring point,
    then slice.  With ignore_hemispheres False the cap halves of an icosphere come first (upper, then lower)."""
    parts = []
    if not ignore_hemispheres:
        sph = get_sphere_point_surface_samples(capsule_radius, device, num_subdivisions=num_sphere_subdivisons)
        up = sph[sph[:, 2] > 1e-5].clone()
        up[:, 2] += capsule_length / 2.0
        lo = sph[sph[:, 2] < -1e-5].clone()
        lo[:, 2] -= capsule_length / 2.0
        parts += [up, lo]
    z = _unit(num_cylinder_slices, device) * capsule_length - capsule_length / 2.0
    theta = torch.linspace(0, 2 * torch.pi, num_circle_points + 1, device=device)[:-1]
    ring = torch.empty((num_circle_points, num_cylinder_slices, 3), dtype=torch.float32, device=device)
    ring[..., 0] = (capsule_radius * torch.cos(theta))[:, None]
    ring[..., 1] = (capsule_radius * torch.sin(theta))[:, None]
    ring[..., 2] = z[None, :]
    parts.append(ring.reshape(-1, 3))
    return torch.cat(parts, dim=0)


def _rotate_by_axis_angle(axis, angle, pts):
    """Rotate pts [P, 3] by the unit quaternion of (axis, angle): v + w t + q x t with t = 2 q x v."""
    half = angle / 2
    qv = axis / torch.linalg.vector_norm(axis).clamp(min=1e-9) * torch.sin(half)
    q = torch.cat([qv, torch.cos(half).reshape(1)])
    q = q / torch.linalg.vector_norm(q).clamp(min=1e-9)
    qv, qw = q[:3].expand_as(pts), q[3]
    t = 2 * torch.cross(qv, pts, dim=-1)
    return pts + qw * t + torch.cross(qv, t, dim=-1)


def get_char_point_samples(char_model, sphere_num_subdivisions=0, box_num_slices=2, box_dim_x=3, box_dim_y=6, capsule_num_circle_points=4,
                           capsule_num_sphere_subdivisons=0, capsule_num_cylinder_slices=4):
    """-> list over bodies of [P_b, 3] float32 tensors on the model's device (a body without geoms gets its origin)."""
    from ..anim.kin_char_model import GeomType
    target_device = char_model._device
    device = "cpu"          # a few hundred setup points: built on the host so they are the same bits whatever the device's sin / cos
    f32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
    out = []
    for b in range(char_model.get_num_joints()):
        body = []
        for g in char_model.get_geoms(b):
            if g._shape_type == GeomType.SPHERE:
                body.append(get_sphere_point_surface_samples(float(f32(g._dims).reshape(-1)[0].item()), device, sphere_num_subdivisions) + f32(g._offset))
            elif g._shape_type == GeomType.BOX:
                body.append(get_box_point_surface_samples(f32(g._dims), device, num_slices=box_num_slices, dim_x=box_dim_x, dim_y=box_dim_y)
                            + f32(g._offset))
            elif g._shape_type == GeomType.CAPSULE:
                span = f32(g._dims)                                   # end - start
                centre = f32(g._offset) + span / 2.0
                zax = torch.tensor([0.0, 0.0, 1.0], device=device)
                axis = torch.linalg.cross(zax, span)
                axis = zax if torch.linalg.vector_norm(axis) < 1e-5 else axis / torch.linalg.vector_norm(axis)
                # (the reference takes the angle from dot(axis, span), not from dot(z, span): a quarter turn about the
                # horizontal normal of the span, or a turn about z for vertical capsules - restated as is)
                angle = torch.acos(torch.dot(axis, span))
                pts = get_capsule_point_surface_samples(torch.linalg.vector_norm(span).item(), g._radius, device,
                                                        num_cylinder_slices=capsule_num_cylinder_slices,
                                                        num_circle_points=capsule_num_circle_points,
                                                        num_sphere_subdivisons=capsule_num_sphere_subdivisons)
                body.append(_rotate_by_axis_angle(axis, angle, pts) + centre)
            else:
                body.append(torch.zeros((1, 3), dtype=torch.float32, device=device))
        if not body:
            body.append(torch.zeros((1, 3), dtype=torch.float32, device=device))
        out.append(torch.cat(body, dim=0).to(target_device))
    return out


def sdSphere(p, c, r):
    """signed distance of points p to the sphere of centre c and radius r (reference util/geom_util.py:167-171)"""
    return torch.linalg.vector_norm(p - c, dim=-1) - r


def get_xy_grid_points(center, dx, dy, num_x_neg, num_x_pos, num_y_neg, num_y_pos):
    """[1 + x_neg + x_pos, 1 + y_neg + y_pos, 2] grid of xy points around `center`, x index first (reference util/geom_util.py:210-221;
    the coordinates come from linspace evaluated on the host so that they are the same bits on every device)."""
    cx, cy = float(center[0]), float(center[1])
    xs = torch.linspace(cx - dx * num_x_neg, cx + dx * num_x_pos, num_x_neg + num_x_pos + 1)
    ys = torch.linspace(cy - dy * num_y_neg, cy + dy * num_y_pos, num_y_neg + num_y_pos + 1)
    x, y = torch.meshgrid(xs, ys, indexing="ij")
    return torch.stack([x, y], dim=-1).to(center.device)
