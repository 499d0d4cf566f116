"""Triage aid (test infrastructure): distribution of per-triangle gradient differences HIP vs oracle at full size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import synthetic, helpers
P, W, H, D, variant = [int(x) for x in sys.argv[1:6]]
s = synthetic.scene(P, W, H, D, seed=42)
of = helpers.oracle_forward(s, True, False, variant=variant); ob = helpers.oracle_backward(s, of, True)
hf = helpers.hip_forward_backward(s, True, False, variant=variant)
for k in ("out_feature", "depth", "normal", "contrib_sum", "contrib_max"):
    print(f"{k:12s} relL2 {helpers.rel_l2(hf[k], of[k]):.3e}")
for k in ("dL_dvertex", "dL_dcenter2D", "dL_dshs", "dL_dopacity"):
    h, o = hf[k].astype(np.float64).reshape(P, -1), ob[k].astype(np.float64).reshape(P, -1)
    err = np.linalg.norm(h - o, axis=1); own = np.linalg.norm(o, axis=1); ref = np.linalg.norm(own)
    order = np.argsort(-err)
    print(f"{k:12s} relL2 {np.linalg.norm(err)/ref:.3e}; excluding top 1e-4/1e-3/1e-2 of triangles: "
          + " ".join(f"{np.linalg.norm(err[order[int(P*f):]])/ref:.3e}" for f in (1e-4, 1e-3, 1e-2))
          + f"; triangles with err > 1% of own: {(err > 0.01*own).sum()}, > 10%: {(err > 0.1*own).sum()}")
cm_h, cm_o = hf["contrib_max"].astype(np.float64), of["contrib_max"].astype(np.float64)
d = np.abs(cm_h - cm_o); o2 = np.argsort(-d)
print("contrib_max worst", [(int(i), float(cm_h[i]), float(cm_o[i])) for i in o2[:5]], "relL2 w/o top 1e-4:", np.linalg.norm(d[o2[int(P*1e-4):]])/np.linalg.norm(cm_o))
np.savez("gpurun_out/fullsize_worst.npz", worst_v=np.argsort(-np.linalg.norm((hf["dL_dvertex"]-ob["dL_dvertex"]).reshape(P,-1),axis=1))[:2000],
         hv=hf["dL_dvertex"][np.argsort(-np.linalg.norm((hf["dL_dvertex"]-ob["dL_dvertex"]).reshape(P,-1),axis=1))[:2000]], worst_cm=o2[:200], cm_h=cm_h[o2[:200]])
