"""Dumps the per-tile lists of a synthetic scene (from the CPU oracle's forward) for tools/sim/group_sim.c."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import synthetic, helpers
P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = synthetic.scene(P, W, H, 3, seed=42)
of = helpers.oracle_forward(s, rich_info=True)
st = of["state"]
N = of["num_rendered"]
with open(sys.argv[4], "wb") as f:
    np.array([P, N, W, H], np.int32).tofile(f)
    for k in ("v1_2D", "v2_2D", "v3_2D"):
        st.field(k).astype(np.float32).tofile(f)
    s["opacity"].astype(np.float32).ravel().tofile(f)
    st.field("vals").astype(np.uint32).tofile(f)
    st.field("ranges").astype(np.uint32).tofile(f)
print("N", N)
