"""Initialisation of a triangle model from a point cloud -- `VanillaTSModel.create_from_pcd` with its helpers, on explicit tensors
(src/diff_recon/models/VanillaTS_model.py:761-804 `_sample_points`, :830-917 `create_from_pcd`; src/diff_recon/models/model_utils.py:34-57, 95-149
`inter_point_distance`, `get_inside_mask`, `grid_sampling`, `grid_size_search`; src/diff_recon/utils/sh_utils.py:103-104 `RGB2SH`).

This is the one consumer of `simple_knn.distCUDA2` in the reference: every point becomes an EQUILATERAL triangle in the plane normal
to its normal, with circum-radius = the root of the mean squared distance to its three nearest neighbours; with back-face culling every triangle
gets a twin with the opposite winding.  The model class, its config system and its logger are out of scope: the function
takes what they hold as arguments and returns the four parameter tensors under the reference's attribute names.

Random numbers (random normals when the cloud has none, the offsets of duplicated points, random opacities, random sampling) are drawn like the
reference draws them -- torch's GLOBAL generator of the tensors' device unless `generator` is given -- in the reference's order, so that a seeded run
reproduces the reference's stream (tests/golden/create_from_pcd.npz was produced by the reference's method on CPU tensors)."""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch

SH_C0 = 0.28209479177387814  # sh_utils.py:23


def RGB2SH(rgb: torch.Tensor) -> torch.Tensor:
    return (rgb - 0.5) / SH_C0


def inverse_sigmoid(x: torch.Tensor) -> torch.Tensor:  # model_utils.py:7-8
    return torch.log(x / (1 - x))


def inter_point_distance(pc: torch.Tensor) -> torch.Tensor:
    """model_utils.py:34-36: root of the mean squared distance to the three nearest neighbours (exact search: simple_knn.distCUDA2, csrc/knn.hip)."""
    assert pc.dim() == 2 and pc.size(1) == 3
    from simple_knn import distCUDA2
    return distCUDA2(pc).clamp_(min=1e-10).sqrt()


def get_inside_mask(points: torch.Tensor, bbox: Optional[Sequence[float]]) -> torch.Tensor:
    """model_utils.py:39-57."""
    if bbox is None:
        return torch.ones_like(points[:, 0], dtype=torch.bool)
    if len(bbox) == 4:
        x_min, y_min, x_max, y_max = bbox
        return (points[:, 0] >= x_min) & (points[:, 0] <= x_max) & (points[:, 1] >= y_min) & (points[:, 1] <= y_max)
    if len(bbox) == 6:
        x_min, y_min, z_min, x_max, y_max, z_max = bbox
        return ((points[:, 0] >= x_min) & (points[:, 0] <= x_max) & (points[:, 1] >= y_min) & (points[:, 1] <= y_max)
                & (points[:, 2] >= z_min) & (points[:, 2] <= z_max))
    raise ValueError(f"bbox must be of length 4 or 6, but got {len(bbox)}")


def grid_sampling(xyz: torch.Tensor, *attrs: torch.Tensor, grid_size: float = 0.0):
    """model_utils.py:95-119: one point per occupied grid cell (the cell's centre), attributes averaged over the cell."""
    if grid_size == 0.0:
        return xyz if len(attrs) == 0 else (xyz, *attrs)
    grid_coords = torch.round(xyz / grid_size).int()
    if len(attrs) == 0:
        return torch.unique(grid_coords, dim=0).float() * grid_size
    unique, inverse = torch.unique(grid_coords, return_inverse=True, dim=0)
    sampled_xyz = unique.float() * grid_size
    out = []
    for attr in attrs:
        acc = torch.zeros((sampled_xyz.shape[0], attr.shape[1]), dtype=torch.float32, device=attr.device)
        acc.scatter_reduce_(0, inverse.unsqueeze(1).expand(-1, attr.shape[1]), attr, "mean")  # include_self=True like the reference's call
        out.append(acc)
    return (sampled_xyz, *out)


def grid_size_search(xyz: torch.Tensor, n_sample: Optional[int], tolerance: float = 0.1, max_retry: int = 10) -> float:
    """model_utils.py:122-149: bisection on the cell size until the number of occupied cells is within `tolerance` of n_sample."""
    if n_sample is None or n_sample >= xyz.shape[0]:
        return 0.0
    lo, hi = 0.0, (xyz.max(dim=0).values - xyz.min(dim=0).values).max().item()
    n_min, n_max = n_sample - tolerance * n_sample, n_sample + tolerance * n_sample
    grid_size = hi / n_sample ** (1 / 3)
    for _ in range(max_retry):
        n = grid_sampling(xyz, grid_size=grid_size).shape[0]
        if n_min <= n <= n_max:
            return grid_size
        if n < n_min:
            hi = grid_size
        else:
            lo = grid_size
        grid_size = (lo + hi) / 2
    return grid_size


def sample_points(points: torch.Tensor, shs: torch.Tensor, normals: torch.Tensor, *, sample_method: str = "direct", n_sample: Optional[int] = None,
                  grid_size: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """VanillaTS_model.py:761-804 for one of its two groups (inside / outside the scene box)."""
    if sample_method == "random":
        if n_sample is None or n_sample > points.shape[0] or n_sample <= 0:
            return points, shs, normals  # "target sample number is invalid, using all points"
        idx = torch.randperm(points.shape[0])[:n_sample]  # the reference draws on the CPU generator (:789)
        idx = idx.to(points.device)
        return points[idx], shs[idx], normals[idx]
    if sample_method == "grid":
        g = grid_size_search(points, n_sample) if grid_size is None else grid_size
        p, s, n = grid_sampling(points, shs, normals, grid_size=g)
        return p, s, n / n.norm(dim=1, keepdim=True)
    if sample_method == "direct":
        return points, shs, normals
    raise ValueError(f"Unknown sampling method: {sample_method}")


def create_from_pcd(points: torch.Tensor, colors: torch.Tensor, normals: Optional[torch.Tensor] = None, *, max_sh_degree: int = 3,
                    init_opacity=0.1, duplicate_count: int = 1, back_culling: bool = False, scene_bbox: Optional[Sequence[float]] = None,
                    sample_method: str = "direct", n_sample_inside: Optional[int] = None, n_sample_outside: Optional[int] = None,
                    grid_size_inside: Optional[float] = None, grid_size_outside: Optional[float] = None,
                    device=None) -> Dict[str, torch.Tensor]:
    """VanillaTS_model.py:830-917.  points / normals (N, 3), colors (N, 3) in [0, 1] (the `PointCloud` attributes, any float dtype, any device);
    returns {"_vertex" (P, 3, 3), "_opacity" (P, 1) raw (inverse sigmoid), "_f_dc" (P, 1, 3), "_f_rest" (P, (D + 1)^2 - 1, 3)} on `device`
    (default: the HIP device), float32 -- what the reference wraps into nn.Parameters.  The neighbour search runs on the HIP device."""
    dev = torch.device(device) if device is not None else torch.device("cuda")
    points = torch.as_tensor(points).float().to(dev)
    shs = RGB2SH(torch.as_tensor(colors).float().to(dev))
    normals = torch.zeros_like(points) if normals is None else torch.as_tensor(normals).float().to(dev)
    if not normals.any():
        normals = torch.randn_like(points)  # :849-850
    normals = normals / normals.norm(dim=1, keepdim=True)

    inside = get_inside_mask(points, scene_bbox)
    groups = []
    for mask, n_sample, grid in ((inside, n_sample_inside, grid_size_inside), (~inside, n_sample_outside, grid_size_outside)):
        groups.append(sample_points(points[mask], shs[mask], normals[mask], sample_method=sample_method, n_sample=n_sample, grid_size=grid))
    points = torch.cat((groups[0][0], groups[1][0]), dim=0)
    shs = torch.cat((groups[0][1], groups[1][1]), dim=0)
    normals = torch.cat((groups[0][2], groups[1][2]), dim=0)
    scaling = inter_point_distance(points)[..., None]

    n = points.shape[0]
    if init_opacity == "random":
        opacities = inverse_sigmoid(torch.rand((n, 1)).float().to(dev))
    else:
        opacities = inverse_sigmoid(torch.ones((n, 1)).float().to(dev) * init_opacity)
    features = torch.zeros((n, (max_sh_degree + 1) ** 2, 3), dtype=torch.float32, device=dev)
    features[:, 0, :] = shs

    if duplicate_count > 1:  # :877-889: jittered copies inside half the neighbour distance, sizes recomputed on the denser cloud
        copies = [points]
        for _ in range(duplicate_count - 1):
            off = torch.rand((scaling.shape[0], 3)).float().to(dev)  # the reference draws on the CPU generator and moves the result
            copies.append(points + (off * 2 - 1) * 0.5 * scaling)
        points = torch.cat(copies, dim=0)
        opacities = opacities.repeat(duplicate_count, 1)
        features = features.repeat(duplicate_count, 1, 1)
        normals = normals.repeat(duplicate_count, 1)
        scaling = inter_point_distance(points)[..., None]

    # equilateral triangles in the plane normal to `normals` (:894-906)
    up = torch.tensor([0.0, 0.0, 1.0], device=dev).repeat(points.shape[0], 1)
    u_dir = torch.cross(up, normals, dim=1)
    u_dir[u_dir.norm(dim=1) < 1e-10] = torch.tensor([1.0, 0.0, 0.0], device=dev)
    u_dir = u_dir / u_dir.norm(dim=1, keepdim=True)
    v_dir = torch.cross(normals, u_dir, dim=1)
    v_dir[v_dir.norm(dim=1) < 1e-10] = torch.tensor([0.0, 1.0, 0.0], device=dev)
    v_dir = v_dir / v_dir.norm(dim=1, keepdim=True)
    v1 = points + u_dir * scaling
    v2 = points + (-1 / 2 * u_dir + math.sqrt(3) / 2 * v_dir) * scaling
    v3 = points + (-1 / 2 * u_dir - math.sqrt(3) / 2 * v_dir) * scaling
    vertex = torch.stack((v1, v2, v3), dim=1)
    if back_culling:  # :908-912: the same triangles with the opposite winding
        vertex = torch.cat((vertex, torch.stack((v3, v2, v1), dim=1)), dim=0)
        opacities = torch.cat((opacities, opacities), dim=0)
        features = torch.cat((features, features), dim=0)
    return {"_vertex": vertex.contiguous(), "_opacity": opacities.contiguous(), "_f_dc": features[:, :1].contiguous(), "_f_rest": features[:, 1:].contiguous()}
