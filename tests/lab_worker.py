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

out = []
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
