"""Times render_fwd / render_bwd of the headline scene (HIP events of libts2d's profile hook, 30 launches after 5 warm-up ones)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
from diff_triangle_rasterization_2D import _C
P, W, H, D = int(os.environ.get("TS_P", 1_000_000)), int(os.environ.get("TS_W", 1920)), int(os.environ.get("TS_H", 1080)), int(os.environ.get("TS_D", 3))
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s = synthetic.scene(P, W, H, D, seed=42)
for _ in range(5):
    helpers.hip_forward_backward(s, rich_info=True, variant=variant)
torch.cuda.synchronize()
_C.profile_reset(); _C.profile_only(""); _C.profile_enable(True)
for _ in range(30):
    helpers.hip_forward_backward(s, rich_info=True, variant=variant)
torch.cuda.synchronize()
rows = {n: ms / max(k, 1) for n, ms, k in _C.profile_read()}
_C.profile_enable(False)
print(f"P={P} {W}x{H}", os.environ.get("TS2D_BLEND", "default"), " ".join(f"{n} {v:.4f}" for n, v in rows.items() if n.startswith("render")), f"total {sum(rows.values()):.4f}")
