import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, helpers, test_fuzz_gpu as F
s, variant, rich, back, use_feature = F._case(93)
hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
builds = helpers.ref3d_builds(s, rich, back, use_feature, fuzz_seed=93)
of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=3)
names = list(builds)
def bad(x, y):
    scale = np.maximum(np.abs(y), np.median(np.abs(y)))
    return int((~(np.abs(x - y) <= 1e-3 * scale)).sum())
print("size", hf["depth"].size)
for i, a in enumerate(names):
    for b in names[i + 1:]:
        print(a, "vs", b, bad(builds[a]["depth"], builds[b]["depth"]))
for b in names:
    print("product vs", b, bad(hf["depth"], builds[b]["depth"]), " oracle vs", b, bad(of["depth"], builds[b]["depth"]))
print("product vs oracle", bad(hf["depth"], of["depth"]))
