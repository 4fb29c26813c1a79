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
