"""Times the fused depth / normal consistency loss (fwd + bwd) at one size; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table.
    python tools/bench_depth_normal.py [H W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]
import torch
from diff_recon_hip import DepthNormalLoss
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 800)
g = torch.Generator(device="cuda").manual_seed(0)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H, device="cuda"), torch.linspace(0, 1, W, device="cuda"), indexing="ij")
depth = (5 + 2 * torch.sin(3 * xx + 1) * torch.cos(2 * yy) + 0.05 * torch.randn((H, W), device="cuda", generator=g)).requires_grad_(True)
normal = (torch.randn((3, H, W), device="cuda", generator=g) * 0.3 + torch.tensor([0.1, -0.2, -1.0], device="cuda")[:, None, None]).requires_grad_(True)
loss = DepthNormalLoss(scale_factor=0.5)
def step():
    depth.grad = normal.grad = None
    loss(depth, normal, 0.3, 0.3).backward()
for _ in range(10): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print(f"{H}x{W}: {(time.perf_counter() - t) / 50 * 1e3:.4f} ms per fwd+bwd (eager, incl. host)")
