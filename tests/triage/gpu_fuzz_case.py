"""Triage aid (test infrastructure): per-triangle error distribution of one fuzz case (tests/test_fuzz_gpu.py:_case)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import helpers, test_fuzz_gpu as F
seed = int(sys.argv[1])
s, variant, rich, back, uf = F._case(seed)
print("variant", variant, "rich", rich, "back", back, "feature", uf, "gamma", s["gamma"], "WxH", s["image_width"], s["image_height"], "P", len(s["vertex"]), "D", s["sh_degree"])
of = helpers.oracle_forward(s, rich, back, use_feature=uf, variant=variant); ob = helpers.oracle_backward(s, of, rich, use_feature=uf)
hf = helpers.hip_forward_backward(s, rich, back, use_feature=uf, variant=variant)
P = len(s["vertex"])
for k in ("dL_dvertex", "dL_dcenter2D", "dL_dopacity"):
    err = np.linalg.norm((hf[k].astype(np.float64) - ob[k]).reshape(P, -1), axis=1); own = np.linalg.norm(ob[k].astype(np.float64).reshape(P, -1), axis=1); ref = np.linalg.norm(own)
    o = np.argsort(-err)[:4]
    print(k, "relL2", np.linalg.norm(err) / ref, "worst", [(int(i), float(err[i] / ref), float(err[i] / max(own[i], 1e-30))) for i in o], "w/o worst", np.linalg.norm(np.delete(err, o[:1])) / ref)
nc_h = helpers.hip_state(hf, s, "n_contrib").astype(np.int64); nc_o = of["state"].field("n_contrib").astype(np.int64)
print("n_contrib mismatches", int((nc_h != nc_o).sum()), "image relL2", helpers.rel_l2(hf["out_feature"], of["out_feature"]))
