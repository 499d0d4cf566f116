"""3D variant: which triangles carry the dL_dvertex distance between the HIP path (H), the oracle (O) and the reference build (R)?
    python tests/triage/worst_triangles3d.py [P W H D [seed]]  -> gpurun_out/worst_triangles3d_<P>.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import helpers  # noqa: E402
import ref3d_f64  # noqa: E402
import ref_build  # noqa: E402
import synthetic  # noqa: E402
import test_parity3d_gpu as T3  # noqa: E402


def main():
    P, W, H, D = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (93000, 1600, 1600, 0)))
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 42
    s = synthetic.scene(P, W, H, D, seed=seed)
    of = helpers.oracle_forward(s, True, False, variant=3)
    ob = helpers.oracle_backward(s, of, True)
    hf = helpers.hip_forward_backward(s, True, False, variant=3)
    R = ref_build.forward_backward(s, True, False, variant=3)
    st = of["state"]
    g = {"H": hf["dL_dvertex"].astype(np.float64), "O": ob["dL_dvertex"].astype(np.float64), "R": R["dL_dvertex"].astype(np.float64)}
    tot = np.linalg.norm(g["R"])
    mag = np.linalg.norm(g["R"].reshape(P, -1), axis=1)
    err = {k: np.linalg.norm((g[k[0]] - g[k[1]]).reshape(P, -1), axis=1) for k in ("HR", "OR", "HO")}
    lines = [f"scene S(P={P}, {W}x{H}, D={D}, seed={seed}) 3D; |dL_dvertex|_R = {tot:.4e}; rel-L2: " +
             ", ".join(f"{k} {np.linalg.norm(e) / tot:.3e}" for k, e in err.items())]
    for pair in ("HR", "OR"):
        order = np.argsort(-err[pair])
        lines.append(f"--- worst by {pair}")
        for r in range(15):
            i = int(order[r])
            v = [st.field(f"v{k}_view")[i].astype(np.float64) for k in (1, 2, 3)]
            edges = [np.linalg.norm(v[1] - v[0]), np.linalg.norm(v[2] - v[1]), np.linalg.norm(v[0] - v[2])]
            lines.append(f"#{r:2d} id={i} HR={err['HR'][i]:.3e} OR={err['OR'][i]:.3e} HO={err['HO'][i]:.3e} ({err[pair][i] / tot:.2e} of total) |g|={mag[i]:.3e} "
                         f"cos={T3._grazing_cos(st, i):.4f} tie_gap={ref3d_f64.min_tie_gap(s, st, i):.2e} depth={v[0][2]:.1f} edges={edges[0]:.2f}/{edges[1]:.2f}/{edges[2]:.2f} "
                         f"op={s['opacity'][i, 0]:.3f} tiles={int(st.field('tiles_touched')[i])}")
    out = "\n".join(lines)
    print(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"worst_triangles3d_{P}.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
