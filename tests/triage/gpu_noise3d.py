"""Triage aid (test infrastructure: compares against the oracle, hence under tests/).  Scratch (GPU box): fp32 noise of oracle vs HIP for the 3D variant, against the float64 autograd model."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import synthetic, helpers, ref3d_f64

def view_to_c2d(s, gv):
    view = s["viewmatrix"].astype(np.float64)
    return (gv.sum(1) @ view[:3, :3])[:, :2]

for (P, W, H, seed, edge) in [(300, 64, 64, 1, 6.0), (1500, 96, 80, 2, 4.0)]:
    s = synthetic.scene(P, W, H, 1, seed=seed, edge_px=edge)
    s["opacity"] = np.ones_like(s["opacity"])
    of = helpers.oracle_forward(s, variant=3); ob = helpers.oracle_backward(s, of)
    hf = helpers.hip_forward_backward(s, variant=3)
    st = of["state"]
    order = np.argsort(st.field("depth"), kind="stable")
    pairs = ref3d_f64.processed_pairs(st, P, W, H)
    img, dep, nor, gv, gsh, gop = ref3d_f64.loss_and_grads(s, 1, order, pairs)
    c2d = view_to_c2d(s, gv)
    print(f"P={P} {W}x{H}: N={of['num_rendered']}")
    for name, truth, o, h in [("image", img, of["out_feature"], hf["out_feature"]), ("depth", dep, of["depth"], hf["depth"]),
                              ("normal", nor, of["normal"], hf["normal"]), ("dL_dvertex", gv, ob["dL_dvertex"], hf["dL_dvertex"]),
                              ("dL_dshs", gsh, ob["dL_dshs"], hf["dL_dshs"]), ("dL_dcenter2D", c2d, ob["dL_dcenter2D"], hf["dL_dcenter2D"])]:
        print(f"  {name:13s} oracle-vs-f64 {helpers.rel_l2(o, truth):.3e}  hip-vs-f64 {helpers.rel_l2(h, truth):.3e}  hip-vs-oracle {helpers.rel_l2(h, o):.3e}")
