"""Triage of the HIP-graph replay (round 5): which variants of the capture recipe give finite, correct gradients."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import torch
import helpers, synthetic
import diff_triangle_rasterization_2D as pkg
from diff_triangle_rasterization_2D import TriangleRasterizer as R2

P = 12000
a = synthetic.scene(P, 256, 192, 2, seed=61)
b = synthetic.scene(P, 256, 192, 2, seed=62)
want = helpers.hip_forward_backward(b, True)
cap = 2 * max(want["num_rendered"], helpers.hip_forward_backward(a, True, backward=False)["num_rendered"]) + 4096
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()


def run(clear_keep, keep_out, zero_grads_in_fn):
    vertex, shs, opacity = t(a["vertex"]).requires_grad_(True), t(a["shs"]).requires_grad_(True), t(a["opacity"]).requires_grad_(True)
    gi, gd, gn = t(a["dL_dout_feature"]), t(a["dL_dout_depth"]), t(a["dL_dout_normal"])
    raster = R2(helpers.hip_settings(a, True))
    keep = {}

    def step():
        if clear_keep:
            keep.clear()
        if zero_grads_in_fn:
            vertex.grad = shs.grad = opacity.grad = None
        c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
        out = raster(vertex, c2d, opacity, shs=shs)
        torch.autograd.backward([out[0], out[2], out[3]], [gi, gd, gn])
        if keep_out:
            keep.update(out=out, c2d=c2d)

    pkg.set_instance_capacity(cap)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                vertex.grad = shs.grad = opacity.grad = None
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        vertex.grad = shs.grad = opacity.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        torch.cuda.synchronize()
        ptr0 = vertex.grad.data_ptr()
        with torch.no_grad():
            vertex.copy_(t(b["vertex"])); shs.copy_(t(b["shs"])); opacity.copy_(t(b["opacity"]))
            gi.copy_(t(b["dL_dout_feature"])); gd.copy_(t(b["dL_dout_depth"])); gn.copy_(t(b["dL_dout_normal"]))
        res = []
        for rep in range(3):
            graph.replay()
            torch.cuda.synchronize()
            g = vertex.grad.detach().cpu().numpy()
            res.append((int(np.isnan(g).sum()), float(helpers.rel_l2(np.nan_to_num(g), want["dL_dvertex"])), vertex.grad.data_ptr() == ptr0))
        return res
    finally:
        pkg.set_instance_capacity(None)


for clear_keep in (False, True):
    for keep_out in (True, False):
        for zero in (False, True):
            try:
                print("clear_keep", clear_keep, "keep_out", keep_out, "zero_in_fn", zero, "->", run(clear_keep, keep_out, zero), flush=True)
            except Exception as e:
                print("clear_keep", clear_keep, "keep_out", keep_out, "zero_in_fn", zero, "EXC", repr(e)[:300], flush=True)
