"""Integer state (num_rendered, radii) of the 3D variant on extended fuzz seeds: product / oracle / the reference's three builds."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, helpers, test_fuzz_gpu as F
for seed in [int(x) for x in sys.argv[1:]]:
    s, variant, rich, back, use_feature = F._case(seed)
    hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant, backward=False)
    of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=3)
    builds = helpers.ref3d_builds(s, rich, back, use_feature, fuzz_seed=seed)
    print("seed", seed, "W H P", s["image_width"], s["image_height"], s["vertex"].shape[0], "gamma", s["gamma"], "back", back, "N hip/oracle", hf["num_rendered"], of["num_rendered"])
    if builds is None:
        print("  reference died"); continue
    for b, o in builds.items():
        d = np.nonzero(o["radii"] != hf["radii"])[0]
        print("  ", b, "N", o["num_rendered"], "radii differ at", d[:8], "ref", o["radii"][d[:8]], "hip", hf["radii"][d[:8]])
        for i in d[:3]:
            print("     tri", i, "vertex", s["vertex"][i].tolist())
