"""Runs parity cases against the oracle in a SEPARATE process that loads the lab library (tools/bin/libts2d_lab.so: the product's
objects + the measurement kernels of earlier rounds, selected by TS2D_BLEND / TS2D_BWD).  The product library has none of them.

    TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so TS2D_BWD=mfma python tests/lab_worker.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd"), os.path.join(ROOT, "tests")]
import helpers  # noqa: E402
import synthetic  # noqa: E402

import numpy as np  # noqa: E402

FORCE_TICKETS = os.environ.get("LAB_FORCE_TICKETS") == "1"
if FORCE_TICKETS:
    # every sort and scan takes the hierarchical (ticket) passes of very large scenes, the ticket-path depth census included
    from diff_triangle_rasterization_2D import _C
    _C._lib.ts2d_lab_force_ticket_passes(1)

out = []
if FORCE_TICKETS:
    for near in (False, True):  # False: every depth shares its top key byte (4th depth pass skipped); True: it varies
        s = synthetic.scene(20000, 200, 120, 2, seed=91)
        if near:
            s["vertex"][: s["vertex"].shape[0] // 2, :, 2] += 900.0  # half of the triangles at a tenth of the distance
        of = helpers.oracle_forward(s, True)
        ob = helpers.oracle_backward(s, of, True)
        hf = helpers.hip_forward_backward(s, True)
        st = of["state"]
        e = {"int_num_rendered": float(hf["num_rendered"] != of["num_rendered"]),
             "int_radii": float(not np.array_equal(hf["radii"], of["radii"])),
             "int_keys": float(not np.array_equal(helpers.hip_state(hf, s, "keys").reshape(-1), st.field("keys").view(np.int64).reshape(-1))),
             "int_vals": float(not np.array_equal(helpers.hip_state(hf, s, "vals").astype(np.int64).reshape(-1), st.field("vals").astype(np.int64).reshape(-1))),
             "image": helpers.rel_l2(hf["out_feature"], of["out_feature"])}
        for k in ("dL_dvertex", "dL_dcenter2D", "dL_dopacity", "dL_dshs"):
            e[k] = helpers.rel_l2(hf[k], ob[k])
        out.append(e)
    print("LAB_RESULT " + json.dumps(out))
    sys.exit(0)
for P, W, H, D, rich, gamma in [(3000, 130, 70, 3, True, 1.0), (4000, 96, 96, 1, False, 2.0)]:
    s = synthetic.scene(P, W, H, D, seed=77)
    s["gamma"] = gamma
    of = helpers.oracle_forward(s, rich)
    ob = helpers.oracle_backward(s, of, rich)
    hf = helpers.hip_forward_backward(s, rich)
    e = {"image": helpers.rel_l2(hf["out_feature"], of["out_feature"])}
    for k in ("dL_dvertex", "dL_dcenter2D", "dL_dopacity", "dL_dshs"):
        e[k] = helpers.rel_l2(hf[k], ob[k])
    if rich:
        for k in ("depth", "normal", "contrib_sum", "contrib_max"):
            e[k] = helpers.rel_l2(hf[k], of[k])
    out.append(e)
print("LAB_RESULT " + json.dumps(out))
