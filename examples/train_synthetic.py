"""End-to-end training loop on a synthetic target, using only this repository's drop-in pieces the way the reference's
trainer uses its own (src/diff_recon/trainers/VanillaTS_trainer.py:60-130, src/diff_recon/models/VanillaTS_model.py:585-694):

    render_view (argument construction of VanillaTSModel.forward)  ->  TriangleRenderer  ->  2D or 3D HIP rasterizer
    photometric_loss (fused L1 + SSIM)  ->  backward through the rasterizer  ->  Adam  ->  DensificationStats.update

A "ground truth" image is rendered from a hidden set of triangles; a perturbed copy is optimised towards it.
    python examples/train_synthetic.py [--rasterizer 2D|3D] [--iters 200] [--triangles 20000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]

import numpy as np
import torch

import synthetic
from diff_recon_hip import DensificationStats, photometric_loss, render_view


class Camera:
    """The attributes of the reference's Camera that the renderer reads (src/diff_recon/utils/camera.py:70-117)."""

    def __init__(self, s, device):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.image_width, self.image_height = s["image_width"], s["image_height"]
        self.tan_fovx, self.tan_fovy = s["tanfovx"], s["tanfovy"]
        self.world_view_transform, self.full_proj_transform, self.camera_center = t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"])
        self.device = device


def train(rasterizer="2D", iters=200, triangles=20000, width=256, height=192, seed=0, log=print):
    dev = torch.device("cuda")
    s = synthetic.scene(triangles, width, height, 1, seed=seed, edge_px=10.0)
    cam = Camera(s, dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bg = torch.zeros(3)
    kw = dict(bg_color=bg, gamma=1.0, active_sh_degree=1, max_sh_degree=1, rasterizer_type=rasterizer)
    with torch.no_grad():  # hidden target
        gt = render_view(cam, t(s["vertex"]), t(s["shs"][:, :1]), t(s["shs"][:, 1:]), torch.logit(t(s["opacity"]).clamp(0.05, 0.95)),
                         is_training=False, **kw)["render"].clamp(0, 1)
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    vertex = (t(s["vertex"]) + 1.5 * torch.randn(s["vertex"].shape, device=dev, generator=g)).requires_grad_(True)
    f_dc = torch.full_like(t(s["shs"][:, :1]), 0.5).requires_grad_(True)
    f_rest = torch.zeros_like(t(s["shs"][:, 1:])).requires_grad_(True)
    raw_op = torch.zeros((triangles, 1), device=dev).requires_grad_(True)
    opt = torch.optim.Adam([{"params": [vertex], "lr": 0.05}, {"params": [f_dc], "lr": 0.01}, {"params": [f_rest], "lr": 0.0005},
                            {"params": [raw_op], "lr": 0.05}])
    stats = DensificationStats(triangles, dev)
    losses, t0 = [], time.perf_counter()
    for it in range(iters):
        pkg = render_view(cam, vertex, f_dc, f_rest, raw_op, is_training=True, **kw)
        loss = photometric_loss(pkg["render"], gt, 0.8, 0.2)  # w_L1 = 1 - w_ssim, VanillaTS_trainer.py:72,111
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        stats.update(pkg)
        losses.append(loss.item())
        if log and (it % 50 == 0 or it == iters - 1):
            log(f"iter {it:4d}  loss {losses[-1]:.5f}  visible {int((pkg['radii'] > 0).sum())}")
    torch.cuda.synchronize()
    return losses, stats, (time.perf_counter() - t0) / iters


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rasterizer", default="2D", choices=["2D", "3D"])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--triangles", type=int, default=20000)
    a = ap.parse_args()
    losses, stats, sec = train(a.rasterizer, a.iters, a.triangles)
    print(f"{a.rasterizer}: loss {losses[0]:.5f} -> {losses[-1]:.5f} in {a.iters} iterations, {sec * 1e3:.2f} ms/iteration (incl. Python)")
