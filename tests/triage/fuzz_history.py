"""History dependence: run the product on a sequence of fuzz seeds in ONE process and compare the last one's gradients with the oracle.
usage: fuzz_history.py <seed> <seed> ... (the last seed is checked after each prefix length is tried)"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, helpers, test_fuzz_gpu as F
seeds = [int(x) for x in sys.argv[1:]]
last = seeds[-1]
s, variant, rich, back, use_feature = F._case(last)
of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=variant)
ob = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
def check(tag):
    hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
    r = {k: helpers.rel_l2(hf[k], ob[k]) for k in ("dL_dopacity", "dL_dvertex")}
    e = np.abs(hf["dL_dopacity"].astype(np.float64) - ob["dL_dopacity"]).ravel()
    bad = np.nonzero(e > 1e-3 * np.abs(ob["dL_dopacity"]).max())[0]
    print(tag, "img", helpers.rel_l2(hf["out_feature"], of["out_feature"]), r, "bad triangles", len(bad), bad[:10].tolist(), flush=True)
check("fresh")
for p in seeds[:-1]:
    sp, vp, rp, bp, up = F._case(p)
    helpers.hip_forward_backward(sp, rp, bp, use_feature=up, variant=vp)
    check(f"after {p}")
