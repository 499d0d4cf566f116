"""Which triangles carry the geometry-gradient discrepancy between the HIP path and the oracle at full size, and what do
they have in common?   python tests/triage/worst_triangles.py [P W H D [seed]]  -> gpurun_out/worst_triangles_<P>.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import helpers  # noqa: E402
import synthetic  # noqa: E402


def main():
    P, W, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (1_000_000, 1920, 1080, 3)))
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 42
    s = synthetic.scene(P, W, H, D, seed=seed)
    of = helpers.oracle_forward(s, True, False)
    ob = helpers.oracle_backward(s, of, True)
    hf = helpers.hip_forward_backward(s, True, False)
    st = of["state"]
    v = np.concatenate([st.field("v1_2D"), st.field("v2_2D"), st.field("v3_2D")], axis=1).astype(np.float64)  # (P, 6)
    area2 = st.field("area2").astype(np.float64)
    err = np.linalg.norm((hf["dL_dvertex"].astype(np.float64) - ob["dL_dvertex"]).reshape(P, -1), axis=1)
    mag = np.linalg.norm(ob["dL_dvertex"].astype(np.float64).reshape(P, -1), axis=1)
    tot = np.linalg.norm(ob["dL_dvertex"].astype(np.float64))
    order = np.argsort(-err)
    lines = [f"scene S(P={P}, {W}x{H}, D={D}, seed={seed}); |dL_dvertex|_oracle = {tot:.4e}; rel-L2 all = {np.linalg.norm(err) / tot:.3e}"]
    e = lambda a, b: np.linalg.norm(a - b, axis=1)
    for r in range(25):
        i = order[r]
        x = v[i]
        edges = [np.hypot(x[2] - x[0], x[3] - x[1]), np.hypot(x[4] - x[2], x[5] - x[3]), np.hypot(x[0] - x[4], x[1] - x[5])]
        hmin = abs(area2[i]) / max(edges)  # smallest height in pixels
        lines.append(f"#{r:2d} id={i} err={err[i]:.3e} ({err[i] / tot:.2e} of total) |g|={mag[i]:.3e} rel={err[i] / max(mag[i], 1e-30):.2e} "
                     f"area2={area2[i]:.4g} min_height_px={hmin:.3g} edges={edges[0]:.1f}/{edges[1]:.1f}/{edges[2]:.1f} op={s['opacity'][i, 0]:.3f} "
                     f"tiles={int(st.field('tiles_touched')[i])} v=({x[0]:.1f},{x[1]:.1f}) hipg={hf['dL_dvertex'][i].ravel()[:3]} orag={ob['dL_dvertex'][i].ravel()[:3]}")
    # how does the relative per-triangle error correlate with sliverness?
    hmin_all = np.abs(area2) / np.maximum(1e-12, np.maximum(np.maximum(e(v[:, 0:2], v[:, 2:4]), e(v[:, 2:4], v[:, 4:6])), e(v[:, 4:6], v[:, 0:2])))
    vis = mag > 0
    rel = err[vis] / np.maximum(mag[vis], 1e-30)
    for lo, hi in [(0, 0.05), (0.05, 0.2), (0.2, 0.5), (0.5, 1), (1, 2), (2, 1e9)]:
        m = (hmin_all[vis] >= lo) & (hmin_all[vis] < hi)
        if m.any():
            lines.append(f"min height in [{lo},{hi}) px: {int(m.sum())} triangles, median rel err {np.median(rel[m]):.2e}, p99 {np.quantile(rel[m], 0.99):.2e}, "
                         f"max {rel[m].max():.2e}, sum err^2 share {float((err[vis][m] ** 2).sum() / (err ** 2).sum()):.3f}")
    out = "\n".join(lines)
    print(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"worst_triangles_{P}.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
