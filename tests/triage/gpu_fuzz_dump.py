"""Triage aid (test infrastructure): dump the HIP outputs of one fuzz case for offline analysis."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import helpers, test_fuzz_gpu as F
seed = int(sys.argv[1])
s, variant, rich, back, uf = F._case(seed)
hf = helpers.hip_forward_backward(s, rich, back, use_feature=uf, variant=variant)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_{seed}.npz"), **{k: v for k, v in hf.items() if isinstance(v, np.ndarray)},
         n_contrib=helpers.hip_state(hf, s, "n_contrib"), final_T=helpers.hip_state(hf, s, "final_T"))
