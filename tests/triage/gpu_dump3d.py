"""Triage aid (test infrastructure: compares against the oracle, hence under tests/).  Scratch (GPU box): dump HIP 3D outputs of one parity case for offline analysis."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import synthetic, helpers
P, W, H, D = [int(x) for x in sys.argv[1:5]]
s = synthetic.scene(P, W, H, D, seed=4321 + P)
s['gamma'] = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
bc = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
hf = helpers.hip_forward_backward(s, True, bc, variant=3)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/dump3d.npz", **{k: v for k, v in hf.items() if isinstance(v, np.ndarray)})
