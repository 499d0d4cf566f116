"""The flow of tests/test_fuzz_gpu.py without pytest: oracle and product fresh per seed, the reference server asked for 3D seeds; on a
large mismatch both sides are recomputed to see WHICH of them moved.  usage: fuzz_flow.py lo hi [repeat]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [R, R + "/triangle-splatting_amd", R + "/tests"]
import numpy as np, helpers, test_fuzz_gpu as F
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
    for seed in range(lo, hi):
        s, variant, rich, back, use_feature = F._case(seed)
        of = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=variant)
        ob = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
        hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
        if of["num_rendered"] == 0:
            continue
        keys = ["dL_dopacity", "dL_dfeature" if use_feature else "dL_dshs", "dL_dvertex"]
        r = {k: float(helpers.rel_l2(hf[k], ob[k])) for k in keys}
        big = max(r.values()) > 1e-2 or hf["num_rendered"] != of["num_rendered"]
        print(rep, seed, "v", variant, "BAD" if big else "ok", {k: f"{v:.2e}" for k, v in r.items()}, flush=True)
        if big:
            ob2 = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
            of2 = helpers.oracle_forward(s, rich, back, use_feature=use_feature, variant=variant)
            ob3 = helpers.oracle_backward(s, of2, rich, use_feature=use_feature)
            hf2 = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
            for k in keys:
                print("    ", k, "oracle again", float(helpers.rel_l2(ob2[k], ob[k])), "oracle fwd+bwd again", float(helpers.rel_l2(ob3[k], ob[k])),
                      "product again", float(helpers.rel_l2(hf2[k], hf[k])), "product2 vs oracle3", float(helpers.rel_l2(hf2[k], ob3[k])), flush=True)
            e = np.abs(hf["dL_dopacity"].astype(np.float64) - ob["dL_dopacity"]).ravel()
            bad = np.argsort(-e)[:8]
            print("     worst triangles", bad.tolist(), "hip", hf["dL_dopacity"].ravel()[bad].tolist(), "oracle", ob["dL_dopacity"].ravel()[bad].tolist(),
                  "hip2", hf2["dL_dopacity"].ravel()[bad].tolist(), "oracle3", ob3["dL_dopacity"].ravel()[bad].tolist())
            np.savez(os.path.join(R, "gpurun_out", f"flow_bad_{seed}_{rep}.npz"), hip=hf["dL_dopacity"], oracle=ob["dL_dopacity"], hip2=hf2["dL_dopacity"],
                     oracle3=ob3["dL_dopacity"], img=hf["out_feature"], img_o=of["out_feature"])
        if variant == 3 and os.environ.get("FLOW_FAULTER") and seed in (183, 185):
            import subprocess
            rc = subprocess.run([sys.executable, os.path.join(R, "tests", "triage", "gpu_faulter.py")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
            print("   faulting neighbour exit code", rc, flush=True)
            if os.environ.get("FLOW_SCRUB"):
                import torch
                x = torch.empty(int(os.environ["FLOW_SCRUB"]) << 20, dtype=torch.uint8, device="cuda")  # sweep the L2s and the memory-side cache
                x.fill_(1); float(x[::4096].sum()); x.add_(1); torch.cuda.synchronize(); del x
        if variant == 3 and os.environ.get("FLOW_IDLE"):
            import time
            time.sleep(float(os.environ["FLOW_IDLE"]))  # the GPU sits idle as long as a reference request would take
        if variant == 3 and not os.environ.get("FLOW_NO_REF"):
            import time
            t0 = time.time()
            b = helpers.ref3d_builds(s, rich, back, use_feature, fuzz_seed=seed)
            if b is None:
                print("   reference died on seed", seed, "after %.2f s" % (time.time() - t0), flush=True)
                time.sleep(float(os.environ.get("FLOW_SLEEP", "0")))
