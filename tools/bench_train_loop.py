"""ms per training iteration of examples/train_synthetic.py's loop (render_view -> fused L1 + SSIM -> backward -> Adam -> per-iteration
statistics; no periodic model updates) at a given size, with round 4's two caller-side changes switchable:
    --bg-depth-float   the reference's behaviour: background_depth converted to a host float in every forward (one device synchronisation)
    --torch-adam       torch.optim.Adam instead of diff_recon_hip.FusedAdam
python tools/bench_train_loop.py [--triangles 300000 --width 800 --height 800 --iters 60]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "examples")]
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--triangles", type=int, default=300_000)
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--height", type=int, default=800)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--rasterizer", default="2D")
ap.add_argument("--bg-depth-float", action="store_true")
ap.add_argument("--torch-adam", action="store_true")
ap.add_argument("--single-sh-tensor", action="store_true", help="one (P, M, 3) colour parameter (lr / lr_tail) instead of f_dc + f_rest: no cat per forward")
a = ap.parse_args()

import diff_recon_hip as D
import train_synthetic as T

if a.bg_depth_float:
    import diff_recon_hip.model_forward as MF
    orig = MF.TriangleRenderer

    def renderer(cam, bg_depth=5000.0, **kw):
        return orig(cam, bg_depth=float(bg_depth), **kw)  # what pybind does to the reference's 0-dim tensor
    MF.TriangleRenderer = renderer
if a.torch_adam:
    D.FusedAdam = torch.optim.Adam

for warm in (True, False):
    losses, m, sec = T.train(a.rasterizer, 8 if warm else a.iters, a.triangles, a.width, a.height, views=2, views_per_step=1, log=None, updates=False,
                             single_sh=a.single_sh_tensor)
print(f"triangles {a.triangles} {a.width}x{a.height} {a.rasterizer} bg_depth={'float (sync)' if a.bg_depth_float else 'device tensor'} "
      f"adam={'torch' if a.torch_adam else 'fused'} colour={'one tensor' if a.single_sh_tensor else 'f_dc + f_rest'}: {sec * 1e3:.3f} ms/iteration, loss {losses[0]:.4f} -> {losses[-1]:.4f}")
