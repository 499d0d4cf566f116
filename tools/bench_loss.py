"""Times the fused photometric loss (fwd+bwd) on MI355X, next to an eager-torch statement of the same loss
(depthwise F.conv2d x5 + autograd: the kernel sequence the reference runs, trainer_utils.py:9-103).  Prints one JSON line.
Algorithmic bytes: forward reads image+gt and writes three derivative maps, backward reads those five and writes the
gradient: 11 floats per element."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]
import numpy as np, torch
import torch.nn.functional as F
from diff_recon_hip import photometric_loss
from diff_triangle_rasterization_2D import _C

def eager_loss(x, g, w1, ws, kernel):
    win = lambda t: F.conv2d(t[None], kernel, padding=5, groups=t.shape[0])[0]
    mu1, mu2 = win(x), win(g)
    s1, s2, s12 = win(x * x) - mu1 * mu1, win(g * g) - mu2 * mu2, win(x * g) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return w1 * (x - g).abs().mean() + ws * (1 - m.mean())

def main():
    C, H, W = 3, int(sys.argv[1]) if len(sys.argv) > 1 else 1080, int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    gen = torch.Generator(device="cuda").manual_seed(0)
    gt = torch.rand((C, H, W), device="cuda", generator=gen)
    img = (0.7 * gt + 0.3 * torch.rand((C, H, W), device="cuda", generator=gen)).requires_grad_(True)
    ax = torch.arange(11, dtype=torch.float32, device="cuda") - 5
    k1 = torch.exp(-ax * ax / (2 * 1.5 ** 2)); k2 = torch.outer(k1, k1); k2 = (k2 / k2.sum())[None, None].repeat(C, 1, 1, 1)

    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

    def fused():
        img.grad = None; photometric_loss(img, gt, 0.8, 0.2).backward()
    def eager():
        img.grad = None; eager_loss(img, gt, 0.8, 0.2, k2).backward()
    fused(); gf = img.grad.clone(); eager(); ge = img.grad.clone()
    _C.profile_reset(); _C.profile_only(""); _C.profile_enable(True)
    ms_fused = timed(fused)
    rows = {n: ms / max(c, 1) for n, ms, c in _C.profile_read()}
    _C.profile_enable(False)
    ms_eager = timed(eager)
    n = C * H * W
    alg = {"photometric_fwd": 5 * 4 * n, "photometric_bwd": 6 * 4 * n}
    print(json.dumps({"workload": f"L1+SSIM fwd+bwd, {C}x{H}x{W} f32", "fused_ms": round(ms_fused, 4), "eager_torch_ms": round(ms_eager, 4),
                      "speedup": round(ms_eager / ms_fused, 1), "kernels_avg_ms": {k: round(v, 4) for k, v in rows.items()},
                      "achieved_GBps": {k: round(alg[k] / (v * 1e-3) / 1e9, 1) for k, v in rows.items() if k in alg},
                      "hbm_frac": {k: round(alg[k] / (v * 1e-3) / 1e9 / 8000, 4) for k, v in rows.items() if k in alg},
                      "grad_rel_l2_fused_vs_eager": float((gf - ge).norm() / ge.norm())}))

if __name__ == "__main__":
    main()
