"""Per-triangle kernel times (HIP events of the library profile hook) for SH degree 3 / 1 / 0, dense and factored SH gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
from diff_triangle_rasterization_2D import _C, parallel, TriangleRasterizer
P, W, H = 1_000_000, 1920, 1080
for D in (3, 1, 0):
    s = synthetic.scene(P, W, H, D, seed=42)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    vertex, shs, opacity = (t(s[k]).requires_grad_(True) for k in ("vertex", "shs", "opacity"))
    rs = helpers.hip_settings(s, True)
    g = [t(s[k]) for k in ("dL_dout_feature", "dL_dout_depth", "dL_dout_normal")]
    for factored in (False, True):
        sink = parallel.ShGradSink()
        def step():
            c2 = torch.zeros((P, 2), device="cuda", requires_grad=True)
            with parallel.factored_sh_grads(sink, enabled=factored):
                out = TriangleRasterizer(rs)(vertex, c2, opacity, shs=shs)
                torch.autograd.backward([out[0], out[2], out[3]], g)
            vertex.grad = shs.grad = opacity.grad = None
            sink.clear() if hasattr(sink, "clear") else None
        step(); torch.cuda.synchronize()
        _C.profile_reset(); _C.profile_only(""); _C.profile_enable(True)
        for _ in range(5): step()
        torch.cuda.synchronize()
        rows = {n: ms / max(k, 1) for n, ms, k in _C.profile_read()}
        _C.profile_enable(False)
        print(f"D={D} M={shs.shape[1]} factored={factored}: preprocess_fwd {rows['preprocess_fwd']*1e3:.1f} us  preprocess_bwd {rows['preprocess_bwd']*1e3:.1f} us")
