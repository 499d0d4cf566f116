"""3D variant: everything the three implementations (H = HIP, O = oracle, R = reference build) say about single triangles.
    python tests/triage/one_triangle3d.py P W H D seed id [id ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import helpers  # noqa: E402
import ref_build  # noqa: E402
import synthetic  # noqa: E402

np.set_printoptions(precision=6, linewidth=200)
P, W, H, D, seed = (int(x) for x in sys.argv[1:6])
ids = [int(x) for x in sys.argv[6:]]
s = synthetic.scene(P, W, H, D, seed=seed)
of = helpers.oracle_forward(s, True, False, variant=3)
ob = helpers.oracle_backward(s, of, True)
hf = helpers.hip_forward_backward(s, True, False, variant=3)
R = ref_build.forward_backward(s, True, False, variant=3)
O = dict(of, **ob)
st = of["state"]
for i in ids:
    print(f"=== triangle {i}: v_view", st.field("v1_view")[i], st.field("v2_view")[i], st.field("v3_view")[i], "n", st.field("normal_view")[i], "op", s["opacity"][i])
    for k in ("contrib_sum", "contrib_max", "dL_dopacity"):
        print(f"  {k:12s} H {np.ravel(hf[k][i])} O {np.ravel(O[k][i])} R {np.ravel(R[k][i])}")
    print("  dL_dshs[0]   H", np.ravel(hf["dL_dshs"][i])[:3], "O", np.ravel(O["dL_dshs"][i])[:3], "R", np.ravel(R["dL_dshs"][i])[:3])
    for nm, g in (("H", hf), ("O", O), ("R", R)):
        print(f"  dL_dvertex {nm}", np.ravel(g["dL_dvertex"][i]))
