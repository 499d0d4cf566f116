"""Triage 2: the pytest flow of test_a_training_step_replays_from_one_hip_graph, variants 2 then 3 in one process, several rounds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import torch
import helpers, synthetic
import diff_triangle_rasterization_2D as pkg
from diff_triangle_rasterization_2D import TriangleRasterizer as R2
from diff_triangle_rasterization_3D import TriangleRasterizer as R3

P = 12000
MODE = sys.argv[1] if len(sys.argv) > 1 else ""
SYNC_AFTER_CAPTURE = "sync" in MODE
NO_CLEAR = "noclear" in MODE       # keep the previous iteration's outputs alive across the step (the variant that passed in call 3)
NO_KEEP = "nokeep" in MODE         # do not keep outputs at all (GraphedStep-like)
WARM2D = "warm" in MODE            # one throw-away capture of a trivial op first
EAGER_FIRST = "eagerfirst" in MODE
LOCAL_GRADS = "localgrads" in MODE  # the upstream gradients are produced INSIDE the step (on the capturing stream) from the static ones
DIRECT = "direct" in MODE           # no autograd: the two native entry points called directly inside the capture  # run the sync-free step eagerly on the DEFAULT stream a few times before the side-stream warm-up


def one(variant):
    a = synthetic.scene(P, 256, 192, 2, seed=61)
    b = synthetic.scene(P, 256, 192, 2, seed=62)
    want = helpers.hip_forward_backward(b, True, variant=variant)
    cap = 2 * max(want["num_rendered"], helpers.hip_forward_backward(a, True, variant=variant, backward=False)["num_rendered"]) + 4096
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    vertex, shs, opacity = t(a["vertex"]).requires_grad_(True), t(a["shs"]).requires_grad_(True), t(a["opacity"]).requires_grad_(True)
    gi, gd, gn = t(a["dL_dout_feature"]), t(a["dL_dout_depth"]), t(a["dL_dout_normal"])
    raster = (R3 if variant == 3 else R2)(helpers.hip_settings(a, True))
    keep = {}

    def step():
        if not NO_CLEAR:
            keep.clear()
        if DIRECT:
            from diff_triangle_rasterization_2D import _C
            rs = raster.raster_settings
            args = (rs.tanfovx, rs.tanfovy, rs.viewmatrix, rs.projmatrix, rs.campos, rs.sh_degree, rs.gamma, rs.scale_modifier, float(rs.background_depth), rs.background)
            empty = torch.Tensor([])
            fw = _C.rasterize_triangles(rs.image_width, rs.image_height, *args, vertex.detach(), shs.detach(), empty, opacity.detach(), False, True, False,
                                        variant=variant, instance_capacity=cap)
            n, img, radii, depth, normal, cs, cm, gb, bb, ib = fw
            gv, gc, gs, gf, go = _C.rasterize_triangles_backward(*args, vertex.detach(), shs.detach(), empty, opacity.detach(), n, radii, gb, bb, ib, gi, gd, gn, True, False,
                                                                 variant=variant)
            keep.update(out=(img, radii, depth, normal, cs, cm), gv=gv)
            return
        c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
        out = raster(vertex, c2d, opacity, shs=shs)
        if LOCAL_GRADS:
            torch.autograd.backward([out[0], out[2], out[3]], [gi * 1.0, gd * 1.0, gn * 1.0])
        else:
            torch.autograd.backward([out[0], out[2], out[3]], [gi, gd, gn])
        if not NO_KEEP:
            keep.update(out=out, c2d=c2d)

    try:
        pkg.set_instance_capacity(cap)
        if EAGER_FIRST:
            for _ in range(2):
                vertex.grad = shs.grad = opacity.grad = None
                step()
            torch.cuda.synchronize()
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
        if SYNC_AFTER_CAPTURE:
            torch.cuda.synchronize()
        with torch.no_grad():
            vertex.copy_(t(b["vertex"])); shs.copy_(t(b["shs"])); opacity.copy_(t(b["opacity"]))
            gi.copy_(t(b["dL_dout_feature"])); gd.copy_(t(b["dL_dout_depth"])); gn.copy_(t(b["dL_dout_normal"]))
        res = []
        for rep in range(2):
            graph.replay()
            torch.cuda.synchronize()
            row = {}
            if DIRECT:
                res.append({"v_err": float(helpers.rel_l2(np.nan_to_num(keep["gv"].cpu().numpy()), want["dL_dvertex"])),
                            "img_err": float(helpers.rel_l2(keep["out"][0].cpu().numpy(), want["out_feature"]))})
                continue
            for k, g in (("v", vertex.grad), ("sh", shs.grad), ("op", opacity.grad)) + (() if NO_KEEP else (("c2d", keep["c2d"].grad), ("img", keep["out"][0]), ("csum", keep["out"][4]))):
                x = g.detach().cpu().numpy()
                row[k] = int((~np.isfinite(x)).sum())
            row["v_err"] = float(helpers.rel_l2(np.nan_to_num(vertex.grad.detach().cpu().numpy()), want["dL_dvertex"]))
            res.append(row)
        return res
    finally:
        pkg.set_instance_capacity(None)


if WARM2D:
    x = torch.zeros(1024, device="cuda")
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0):
        y = x + 1
    g0.replay()
    torch.cuda.synchronize()
for rnd in range(2):
    for variant in (2, 3):
        print("round", rnd, "variant", variant, one(variant), flush=True)
