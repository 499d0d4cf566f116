"""Profiling aid: culling / lane-occupancy statistics of render_fwd on the headline scene.  Needs a library built with
-DTS2D_STATS (TS2D_EXTRA_FLAGS=-DTS2D_STATS python triangle-splatting_amd/build.py --force)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
from diff_triangle_rasterization_2D import _C
P, W, H, D = 1_000_000, 1920, 1080, 3
s = synthetic.scene(P, W, H, D, seed=42)
buf = (ctypes.c_ulonglong * 8)()
_C._lib.ts2d_stats_read(buf, 1)
hf = helpers.hip_forward_backward(s, backward=False)
torch.cuda.synchronize()
_C._lib.ts2d_stats_read(buf, 1)
v = list(buf)
N = hf["num_rendered"]
print(f"N instances {N}; per-quadrant visits {v[0]} ({v[0]/N:.2f} per instance; 4 = no early exit)")
print(f"survive setup cull {v[1]} ({v[1]/v[0]:.3f} of visits); stage-1 hit {v[2]} ({v[2]/v[1]:.3f} of survivors); blended {v[3]} ({v[3]/v[2]:.3f})")
print(f"blended pairs {v[4]}: {v[4]/v[3]:.1f} lanes per blended entry ({v[4]/v[3]/64:.3f} lane occupancy); quadrants {v[5]}")
print(f"lockstep iterations with per-4x4-block lists {v[6]} ({v[6]/v[3]:.3f} of now), per-8x4-half lists {v[7]} ({v[7]/v[3]:.3f})")
