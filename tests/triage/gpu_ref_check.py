"""Triage aid: the REFERENCE's own rasterizer (oracle/_ref/_ref2d_C.so, built from /root/reference by oracle/build_ref.py)
against the oracle and the HIP path on one scene."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
spec = importlib.util.spec_from_file_location("_ref2d_C", os.path.join(ROOT, "oracle", "_ref", "_ref2d_C.so"))
ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
P, W, H, D = [int(x) for x in sys.argv[1:5]] if len(sys.argv) > 4 else (2000, 128, 96, 3)
s = synthetic.scene(P, W, H, D, seed=11)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
args = (W, H, s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), D, 1.0, 1.0, float(s["background_depth"]),
        t(s["background"]), t(s["vertex"]), t(s["shs"]), torch.empty(0, device="cuda"), t(s["opacity"]), False, True, False)
out = ref.rasterize_triangles(*args)
torch.cuda.synchronize()
n, img, radii, depth, normal, csum, cmax, gb, bb, ib = out
of = helpers.oracle_forward(s, True, False)
print("num_rendered ref/oracle", n, of["num_rendered"], "radii equal", np.array_equal(radii.cpu().numpy(), of["radii"]))
for k, v in (("out_feature", img), ("depth", depth), ("normal", normal), ("contrib_sum", csum), ("contrib_max", cmax)):
    print(f"  {k:12s} ref-vs-oracle relL2 {helpers.rel_l2(v.cpu().numpy(), of[k]):.3e}")
bw = ref.rasterize_triangles_backward(s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), D, 1.0, 1.0,
        float(s["background_depth"]), t(s["background"]), t(s["vertex"]), t(s["shs"]), torch.empty(0, device="cuda"), t(s["opacity"]), n, radii, gb, bb, ib,
        t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"]), True, False)
torch.cuda.synchronize()
ob = helpers.oracle_backward(s, of, True)
hf = helpers.hip_forward_backward(s, True, False)
for k, v in zip(("dL_dvertex", "dL_dcenter2D", "dL_dshs", "dL_dfeature", "dL_dopacity"), bw):
    if k == "dL_dfeature": continue
    r = v.cpu().numpy()
    print(f"  {k:12s} ref-vs-oracle {helpers.rel_l2(r, ob[k]):.3e}   hip-vs-ref {helpers.rel_l2(hf[k], r):.3e}   hip-vs-oracle {helpers.rel_l2(hf[k], ob[k]):.3e}")
print("  image hip-vs-ref", helpers.rel_l2(hf["out_feature"], img.cpu().numpy()))
