"""The oracle-only fallback of tests/test_parity3d_gpu.py::_check_outputs on extended fuzz seeds: which quantity misses, by how much, and
whether the float64 evaluation of the same formulas sides with the oracle or sits between the two."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, helpers, test_fuzz_gpu as F
for seed in [int(x) for x in sys.argv[1:]]:
    s, variant, rich, back, use_feature = F._case(seed)
    of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=3)
    ob = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
    hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=3)
    print("seed", seed, "W H P", s["image_width"], s["image_height"], s["vertex"].shape[0], "gamma", s["gamma"], "rich", rich, "feat", use_feature, "N", hf["num_rendered"])
    for k in ["out_feature", "depth", "normal", "contrib_sum", "contrib_max"]:
        if k in hf and k in of and hf[k] is not None and of[k] is not None:
            print("   fwd", k, helpers.rel_l2(hf[k], of[k]))
    for k in ["dL_dopacity", "dL_dfeature" if use_feature else "dL_dshs", "dL_dvertex", "dL_dcenter2D"]:
        e = (hf[k].astype(np.float64) - ob[k]).reshape(len(ob[k]), -1)
        n = np.linalg.norm(e, axis=1)
        top = np.argsort(-n)[:4]
        print("   bwd", k, helpers.rel_l2(hf[k], ob[k]), "worst triangles", top.tolist(), (n[top] / np.linalg.norm(ob[k])).tolist())
