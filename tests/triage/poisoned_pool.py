"""Uninitialised-read hunt: fill torch's caching allocator with a poison pattern before every product call, then compare with the oracle.
usage: poisoned_pool.py <seed> ...   (fuzz seeds of tests/test_fuzz_gpu.py)"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, torch, helpers, test_fuzz_gpu as F

def poison(kind):
    # several block sizes so that small and large requests both land on poisoned memory
    blocks = []
    for n in (1 << 26, 1 << 22, 1 << 22, 1 << 18, 1 << 18, 1 << 14, 1 << 14, 1 << 10, 1 << 10, 256, 256, 256, 256):
        if kind == "nan":
            blocks.append(torch.full((n,), float("nan"), device="cuda"))
        elif kind == "ones":
            blocks.append(torch.full((n,), -1, dtype=torch.int32, device="cuda"))
        else:
            blocks.append(torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    del blocks

for seed in [int(x) for x in sys.argv[1:]]:
    s, variant, rich, back, use_feature = F._case(seed)
    of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=variant)
    ob = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
    base = None
    for kind in ("nan", "ones", "rand", "nan", "rand"):
        poison(kind)
        hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
        keys = ["out_feature", "dL_dopacity", "dL_dvertex", "dL_dfeature" if use_feature else "dL_dshs", "dL_dcenter2D"]
        ref = dict(of); ref.update(ob)
        r = {k: float(helpers.rel_l2(hf[k], ref[k])) for k in keys}
        flag = "BAD" if (hf["num_rendered"] != of["num_rendered"] or not all(v < 2e-3 for v in r.values())) else "ok"
        print(seed, "v", variant, kind, flag, hf["num_rendered"], of["num_rendered"], {k: f"{v:.2e}" for k, v in r.items()}, flush=True)
