"""Triage aid (test infrastructure: compares against the oracle, hence under tests/).  Scratch triage script (GPU box): HIP path vs oracle on a few scenes, prints the diffs."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import synthetic, helpers
from oracle import ts2d_oracle as O

VARIANT = int(os.environ.get('TS_VARIANT', '2'))

def run(P, W, H, D, rich=True, gamma=1.0, **kw):
    s = synthetic.scene(P, W, H, D, **kw); s["gamma"] = gamma
    t = time.time(); of = helpers.oracle_forward(s, rich, variant=VARIANT); ob = helpers.oracle_backward(s, of, rich); to = time.time() - t
    hf = helpers.hip_forward_backward(s, rich, variant=VARIANT)
    print(f"--- P={P} {W}x{H} D={D} rich={rich} gamma={gamma} N(oracle)={of['num_rendered']} N(hip)={hf['num_rendered']} oracle {to:.2f}s")
    st = of["state"]
    for name, oname in [("tiles_touched","tiles_touched"),("vals","vals"),("keys","keys"),("ranges","ranges"),("n_contrib","n_contrib")]:
        a = helpers.hip_state(hf, s, name); b = st.field(oname)
        a = a.astype(np.int64).reshape(-1); b = b.astype(np.int64).reshape(-1)
        if name == "keys": b = st.field("keys").view(np.int64).reshape(-1)
        print(f"  {name:14s} equal={np.array_equal(a,b)} mismatches={(a!=b).sum() if a.shape==b.shape else 'shape'}")
    print("  radii equal", np.array_equal(hf["radii"], of["radii"]))
    print("  final_T maxabs", np.abs(helpers.hip_state(hf, s, "final_T") - st.field("final_T")).max())
    keys = ["out_feature"] + (["depth","normal","contrib_sum","contrib_max"] if rich else [])
    for k in keys: print(f"  {k:12s} relL2 {helpers.rel_l2(hf[k], of[k]):.3e} maxabs {np.abs(hf[k]-of[k]).max():.3e}")
    for k in ["dL_dvertex","dL_dcenter2D","dL_dshs","dL_dopacity"]:
        print(f"  {k:12s} relL2 {helpers.rel_l2(hf[k], ob[k]):.3e} maxabs {np.abs(hf[k]-ob[k]).max():.3e} ref_norm {np.linalg.norm(ob[k]):.3e}")

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(300, 64, 64, 3)
    run(2000, 128, 96, 3)
    run(10000, 256, 256, 0)
    run(10000, 256, 256, 3, rich=False)
    run(5000, 200, 120, 2, gamma=2.5)
    run(1000, 320, 240, 3, mode="maincu")
