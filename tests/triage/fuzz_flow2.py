"""As fuzz_flow.py, but the product is driven here (graph retained): on a mismatch the backward is run again on the SAME forward state,
and the list positions / radii of the wrong triangles are printed.  usage: fuzz_flow2.py lo hi [repeat]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, torch, helpers, test_fuzz_gpu as F

def run(s, rich, back, use_feature, variant):
    if variant == 3:
        from diff_triangle_rasterization_3D import TriangleRasterizer
    else:
        from diff_triangle_rasterization_2D import TriangleRasterizer
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rs = helpers.hip_settings(s, rich, back, False, "cuda")
    vertex = t(s["vertex"]).requires_grad_(True); opacity = t(s["opacity"]).requires_grad_(True)
    center2D = torch.zeros((vertex.shape[0], 2), device="cuda", requires_grad=True)
    shs = feature = None
    if use_feature: feature = t(s["feature"]).requires_grad_(True)
    else: shs = t(s["shs"]).requires_grad_(True)
    out = TriangleRasterizer(rs)(vertex, center2D, opacity, shs=shs, feature=feature)
    loss = (out[0] * t(s["dL_dout_feature"])).sum()
    if rich: loss = loss + (out[2] * t(s["dL_dout_depth"])).sum() + (out[3] * t(s["dL_dout_normal"])).sum()
    leaves = [vertex, opacity]
    def bwd():
        for l in leaves: l.grad = None
        loss.backward(retain_graph=True)
        return {"dL_dvertex": vertex.grad.cpu().numpy().copy(), "dL_dopacity": opacity.grad.cpu().numpy().copy()}
    node = out[0].grad_fn
    return out, node, bwd

lo, hi = int(sys.argv[1]), int(sys.argv[2])
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    for seed in range(lo, hi):
        s, variant, rich, back, use_feature = F._case(seed)
        of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=variant)
        ob = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
        out, node, bwd = run(s, rich, back, use_feature, variant)
        if of["num_rendered"] == 0: continue
        g1 = bwd()
        r = {k: float(helpers.rel_l2(g1[k], ob[k])) for k in g1}
        big = max(r.values()) > 1e-2
        print(rep, seed, "v", variant, "BAD" if big else "ok", r, flush=True)
        if big:
            g2 = bwd(); g3 = bwd()
            print("   backward again on the same state:", {k: float(helpers.rel_l2(g2[k], ob[k])) for k in g2}, {k: float(helpers.rel_l2(g3[k], ob[k])) for k in g3})
            e = np.abs(g1["dL_dopacity"].astype(np.float64) - ob["dL_dopacity"]).ravel()
            bad = np.nonzero(e > 1e-4)[0]
            ev = np.abs(g1["dL_dvertex"].astype(np.float64) - ob["dL_dvertex"]).reshape(len(e), -1).max(1)
            badv = np.nonzero(ev > 1e-3 * np.abs(ob["dL_dvertex"]).max())[0]
            print("   wrong dL_dopacity rows", bad.tolist(), "wrong dL_dvertex rows", badv.tolist())
            radii = out[1].cpu().numpy()
            lo_i, hi_i = max(0, bad.min() - 8), min(len(e), bad.max() + 9)
            print("   radii", lo_i, radii[lo_i:hi_i].tolist())
            saved = node.saved_tensors; gb, bb, ib = saved[5:8]
            P, N, W, H = len(e), node.num_rendered, s["image_width"], s["image_height"]
            vals = helpers.debug_read_state("vals", P, N, W, H, gb, bb, ib).numpy().astype(np.int64).ravel() & 0x0FFFFFFF  # id bits (csrc/ts2d_support.h)
            nc = helpers.debug_read_state("n_contrib", P, N, W, H, gb, bb, ib).numpy().astype(np.int64)
            rg = helpers.debug_read_state("ranges", P, N, W, H, gb, bb, ib).numpy().astype(np.int64)
            print("   N", N, "ranges", rg.ravel().tolist()[:8], "n_contrib max", nc.max(), "min", nc.min())
            print("   list positions of the wrong rows", {int(i): np.nonzero(vals == i)[0].tolist() for i in bad})
            print("   dL_dvertex of wrong rows", g1["dL_dvertex"][bad[:3]].reshape(-1, 9).tolist())
            print("   scratch bytes", 64 * P + 256, "geometry buffer ptr %x binning %x image %x" % (gb.data_ptr(), bb.data_ptr(), ib.data_ptr()))
        del out, node, bwd
