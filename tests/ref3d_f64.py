"""Independent float64 autograd restatement of the 3D variant's forward model (R3D/src/forward.cu:60-306), used to
check the oracle's hand-written 3D backward (CPU test) and to judge fp32 rounding noise of oracle vs HIP (GPU test).

The discrete decisions (depth order, which (triangle, pixel) pairs are blended) are inputs, so the graph holds only
the smooth part -- like the reference's hand-written backward."""
from __future__ import annotations

import numpy as np


def processed_pairs(st, P, W, H):
    """[P, H*W] bool: pair examined by the oracle's forward loop (position in the tile list < n_contrib[pixel])."""
    ranges = st.field("ranges").reshape(-1, 2).astype(np.int64)
    vals = st.field("vals").astype(np.int64)
    ncon = st.field("n_contrib").reshape(H, W).astype(np.int64)
    gx = (W + 15) // 16
    out = np.zeros((P, H * W), bool)
    for tile, (r0, r1) in enumerate(ranges):
        if r1 <= r0:
            continue
        ids = vals[r0:r1]
        tx, ty = tile % gx, tile // gx
        for py in range(ty * 16, min(H, ty * 16 + 16)):
            for px in range(tx * 16, min(W, tx * 16 + 16)):
                out[ids[: ncon[py, px]], py * W + px] = True
    return out


def forward(vertex, shs, opacity, s, D, order, pairs, return_hits=False):
    """vertex (P,3,3), shs (P,M,3), opacity (P,1): float64 torch tensors.  `pairs`: [P, HW] bool candidates (processed
    pairs); the reference's per-pair tests are applied on top (evaluated on detached values)."""
    import torch

    W, H, gamma = s["image_width"], s["image_height"], float(s["gamma"])
    view = torch.tensor(s["viewmatrix"], dtype=torch.float64)
    campos = torch.tensor(s["campos"], dtype=torch.float64)
    P = vertex.shape[0]
    vh = torch.cat([vertex, torch.ones_like(vertex[..., :1])], -1) @ view  # (P,3,4), row-vector convention
    vv = vh[..., :3]
    n = torch.cross(vv[:, 1] - vv[:, 0], vv[:, 2] - vv[:, 0], dim=-1)  # unnormalised (forward.cu:96)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    ray = torch.stack([s["tanfovx"] * (2 * xs.reshape(-1) - W + 1) / W, s["tanfovy"] * (2 * ys.reshape(-1) - H + 1) / H,
                       torch.ones(H * W, dtype=torch.float64)], -1)  # (HW,3)
    den = n @ ray.T  # (P,HW)
    d0 = (vv[:, 0] * n).sum(-1, keepdim=True)
    safe_den = torch.where(den.abs() < 1e-8, torch.ones_like(den), den)
    depth = d0 / safe_den
    p = depth[..., None] * ray[None]  # (P,HW,3)
    p1, p2, p3 = (vv[:, k][:, None, :] - p for k in range(3))
    nn = (n * n).sum(-1, keepdim=True)
    a1 = (torch.cross(p2, p3, dim=-1) * n[:, None, :]).sum(-1) / nn
    a2 = (torch.cross(p3, p1, dim=-1) * n[:, None, :]).sum(-1) / nn
    a3 = 1 - a1 - a2
    ecc = 1 - 3 * torch.minimum(torch.minimum(a1, a2), a3)
    G = torch.exp(-0.5 * ecc.clamp(min=0) ** (2 * gamma))
    alpha = torch.clamp(opacity.reshape(-1, 1) * G, max=0.99)
    with torch.no_grad():
        hits = torch.tensor(pairs) & (den.abs() >= 1e-8) & (ecc >= 0) & (ecc <= 10) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(hits, alpha, torch.zeros_like(alpha))
    # colour (forward.cu:120-128 -> computeColorFromSH at the world-space centre)
    c = vertex.mean(1)
    d = c - campos
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    rgb = C0 * shs[:, 0]
    if D > 0:
        rgb = rgb - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    assert D <= 1
    rgb = torch.clamp(rgb + 0.5, min=0.0)

    o = torch.as_tensor(order)
    alpha_o = alpha[o]
    Tafter = torch.cumprod(1 - alpha_o, 0)
    Tbefore = torch.cat([torch.ones_like(Tafter[:1]), Tafter[:-1]], 0)
    contrib = alpha_o * Tbefore
    bg = torch.tensor(s["background"], dtype=torch.float64)
    img = (contrib[:, None, :] * rgb[o][:, :, None]).sum(0) + Tafter[-1][None] * bg[:, None]
    dep = (contrib * depth[o]).sum(0) + Tafter[-1] * float(s["background_depth"])
    nor = (contrib[:, None, :] * n[o][:, :, None]).sum(0)
    res = (img.reshape(3, H, W), dep.reshape(H, W), nor.reshape(3, H, W))
    return res + (hits,) if return_hits else res


def loss_and_grads(s, D, order, pairs):
    """Runs the model on scene `s` and returns (img, depth, normal, dL_dvertex, dL_dshs, dL_dopacity) as numpy."""
    import torch

    vertex = torch.tensor(s["vertex"], dtype=torch.float64, requires_grad=True)
    shs = torch.tensor(s["shs"], dtype=torch.float64, requires_grad=True)
    opacity = torch.tensor(s["opacity"], dtype=torch.float64, requires_grad=True)
    img, dep, nor = forward(vertex, shs, opacity, s, D, order, pairs)
    loss = (img * torch.tensor(s["dL_dout_feature"], dtype=torch.float64)).sum() \
        + (dep * torch.tensor(s["dL_dout_depth"], dtype=torch.float64)).sum() \
        + (nor * torch.tensor(s["dL_dout_normal"], dtype=torch.float64)).sum()
    loss.backward()
    return (img.detach().numpy(), dep.detach().numpy(), nor.detach().numpy(), vertex.grad.numpy(), shs.grad.numpy(),
            opacity.grad.numpy())


def min_tie_gap(s, st, i):
    """Smallest gap between the two smallest barycentrics of triangle `i` over the pixels the backward can touch
    (0 <= ecc <= 10, G >= 1/255), in float64.  A gap at fp32-rounding level means the reference's argmin choice
    (R3D backward.cu:388-401) -- and with it which vertices receive that pixel's gradient -- is decided by rounding."""
    W, H, gamma = s["image_width"], s["image_height"], float(s["gamma"])
    v = [st.field(f"v{k}_view")[i].astype(np.float64) for k in (1, 2, 3)]
    n = st.field("normal_view")[i].astype(np.float64)
    ys, xs = np.mgrid[0:H, 0:W]
    ray = np.stack([s["tanfovx"] * (2 * xs - W + 1) / W, s["tanfovy"] * (2 * ys - H + 1) / H, np.ones((H, W))], -1)
    den = ray @ n
    den = np.where(np.abs(den) < 1e-8, 1.0, den)
    p = ((v[0] @ n) / den)[..., None] * ray
    p1, p2, p3 = (vk - p for vk in v)
    a1 = (np.cross(p2, p3) @ n) / (n @ n)
    a2 = (np.cross(p3, p1) @ n) / (n @ n)
    a = np.sort(np.stack([a1, a2, 1 - a1 - a2], -1), -1)
    ecc = 1 - 3 * a[..., 0]
    with np.errstate(over="ignore"):
        m = (ecc >= 0) & (ecc <= 10) & (np.exp(-0.5 * np.clip(ecc, 0, None) ** (2 * gamma)) >= 1 / 255)
    return float((a[..., 1] - a[..., 0])[m].min()) if m.any() else np.inf
