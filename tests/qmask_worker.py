"""Runs scenes through the LAB library twice in one process -- quadrant masks as the product forms them, then every quadrant flagged
(ts2d_lab_force_all_quadrants) -- and prints how far apart the results are.  tests/test_qmask_gpu.py asserts on the printed numbers.

    TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so python tests/qmask_worker.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

import helpers  # noqa: E402
import synthetic  # noqa: E402
from diff_triangle_rasterization_2D import _C  # noqa: E402


def grazing(s, frac, rng):
    """Turns a fraction of the triangles edge-on to their own viewing ray (within ~0.5 degrees): the 3D variant's ill-conditioned cases, whose
    plane horizon crosses their tile rectangle (csrc/ts2d_support.h: quad_setup_3d)."""
    v = s["vertex"].copy()
    P = v.shape[0]
    pick = rng.random(P) < frac
    cam = s["campos"].astype(np.float64)
    c = v.mean(1).astype(np.float64)
    ray = c - cam
    ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    e1 = np.cross(ray, rng.normal(size=(P, 3))); e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    size = np.linalg.norm(v[:, 1] - v[:, 0], axis=1)[:, None]
    tilt = rng.normal(0, 0.005, (P, 1))
    along = ray + tilt * np.cross(ray, e1)  # a direction almost along the ray
    new = np.stack([c - 0.5 * size * e1, c + 0.5 * size * e1, c + size * along], 1)
    v[pick] = new[pick].astype(np.float32)
    s = dict(s)
    s["vertex"] = v
    return s


CASES = [
    # variant, P, W, H, D, gamma, kwargs, grazing fraction
    (2, 30000, 320, 240, 2, 1.0, {}, 0.0),
    (2, 20000, 96, 96, 1, 1.0, {"edge_px": 2.0}, 0.0),       # heavy overdraw, early termination
    (2, 4000, 400, 300, 1, 1.0, {"edge_px": 60.0}, 0.0),     # triangles spanning many tiles (long rectangles: the affine step)
    (2, 1000, 320, 240, 3, 1.0, {"mode": "maincu"}, 0.0),    # the reference's main.cu recipe: huge triangles (the unstaged path)
    (2, 20000, 256, 192, 1, 50.0, {}, 0.0),
    (3, 30000, 320, 240, 2, 1.0, {}, 0.0),
    (3, 20000, 320, 240, 1, 1.0, {}, 0.3),                   # 30 % grazing triangles
    (3, 8000, 400, 300, 1, 7.0, {"edge_px": 40.0}, 0.15),
    (3, 20000, 256, 192, 1, 50.0, {}, 0.1),
]
out = []
rng = np.random.default_rng(123)
for variant, P, W, H, D, gamma, kw, gfrac in CASES:
    s = synthetic.scene(P, W, H, D, seed=300 + P + variant, **kw)
    s["gamma"] = gamma
    if gfrac > 0:
        s = grazing(s, gfrac, rng)
    res = []
    for all_quadrants in (0, 1):
        _C._lib.ts2d_lab_force_all_quadrants(all_quadrants)
        hf = helpers.hip_forward_backward(s, True, variant=variant)
        hf["n_contrib"] = helpers.hip_state(hf, s, "n_contrib")
        hf["masks"] = helpers.debug_read_state("vals", s["vertex"].shape[0], hf["num_rendered"], W, H, *hf["buffers"]).numpy().view(np.uint32) >> 28
        res.append(hf)
    _C._lib.ts2d_lab_force_all_quadrants(0)
    a, b = res
    e = {"variant": variant, "P": P, "gamma": gamma, "num_rendered": a["num_rendered"],
         "mask_bits_set_fraction": float(np.unpackbits(a["masks"].astype(np.uint8)).sum() / (4.0 * max(a["masks"].size, 1))),
         "all_quadrants_really_all": bool((b["masks"] == 15).all()),
         "same_num_rendered": a["num_rendered"] == b["num_rendered"]}
    for k in ("out_feature", "depth", "normal", "radii", "n_contrib"):
        e["exact_" + k] = bool(np.array_equal(a[k], b[k]))
        if not e["exact_" + k]:
            e["differing_" + k] = int((a[k] != b[k]).sum())
    for k in ("contrib_sum", "contrib_max", "dL_dvertex", "dL_dcenter2D", "dL_dshs", "dL_dopacity"):
        e[k] = float(helpers.rel_l2(a[k], b[k]))
    out.append(e)
print("QMASK_RESULT " + json.dumps(out))
