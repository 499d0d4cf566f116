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
if os.environ.get("LAB_DEPTH_ORDER") == "1":
    # the one-launch depth order of small scenes (binning.hip: depth_order_small_kernel) has its own census, its own fourth-pass rule and
    # its own block sums.  Same scene through the three forms -- one launch (the product's path up to 12 288 triangles), the multi-launch ticket-free
    # passes (reached through ts2d_lab_force_depth_pass4) and the hierarchical ticket passes (ts2d_lab_force_ticket_passes): the depth permutation,
    # the instance offsets (= block sums + scan), the instance count and the sorted instance list must be IDENTICAL.
    from diff_triangle_rasterization_2D import _C
    _C._lib.ts2d_lab_force_depth_pass4.argtypes = [__import__("ctypes").c_int]
    # (round 6: the one-launch form is used up to 9 216 triangles, the sampled-splitter form above: the sizes straddle both switch-overs)
    for P, culled in [(1, False), (63, False), (64, False), (65, False), (1023, False), (9215, False), (9216, False), (9217, False), (12288, False), (12289, False), (700, True), (9216, True)]:
        s = synthetic.scene(P, 160, 96, 1, seed=500 + P)
        if culled:
            s["vertex"][:, :, 2] += 5000.0  # every triangle behind the camera: all culled, zero instances
        got = {}
        for form, (tick, p4) in {"one_launch": (0, 0), "multi_launch": (0, 1), "tickets": (1, 0)}.items():
            _C._lib.ts2d_lab_force_ticket_passes(tick)
            _C._lib.ts2d_lab_force_depth_pass4(p4)
            hf = helpers.hip_forward_backward(s, True, backward=False)
            got[form] = (int(hf["num_rendered"]), helpers.hip_state(hf, s, "depth_perm").copy(), helpers.hip_state(hf, s, "point_offsets").copy(),
                         helpers.hip_state(hf, s, "vals").copy() if hf["num_rendered"] > 0 else np.zeros(0, np.int32), hf["out_feature"].copy())
        _C._lib.ts2d_lab_force_ticket_passes(0)
        _C._lib.ts2d_lab_force_depth_pass4(0)
        a = got["one_launch"]
        e = {"P": P, "culled": culled, "num_rendered": a[0]}
        for form in ("multi_launch", "tickets"):
            b = got[form]
            # the permutation of CULLED triangles (depth key 0, no instances) is stable in every form too: plain equality
            e[form] = float(a[0] != b[0] or not np.array_equal(a[1], b[1]) or not np.array_equal(a[2], b[2]) or not np.array_equal(a[3], b[3])
                            or not np.array_equal(a[4], b[4]))
        out.append(e)
    print("LAB_RESULT " + json.dumps(out))
    sys.exit(0)
if os.environ.get("LAB_DEPTH_SPLIT") == "1":
    # Round 6: the depth order of mid-sized scenes by sampled splitters + per-bucket sorts in LDS (binning.hip: depth_split_* / depth_bucket_sort_kernel,
    # the product's path between 12 288 and 500 000 triangles; here up to the 1.6 M it supports) against the LSD passes (ts2d_lab_depth_split(1, 0)) and against itself
    # with a per-bucket register capacity of 512 pairs (ts2d_lab_depth_split(2, 512): every ordinary bucket takes the kernel's global-memory
    # path).  Depth permutation, instance offsets (= tiles in depth order + block sums + scan), instance count, sorted instance list and image
    # must be IDENTICAL, on scenes that bend the buckets: a third of the triangles culled (key 0), half of them at ONE depth (a bucket of equal
    # keys larger than the registers: the copy), depths quantised to 40 values (oversize buckets of few distinct keys), a far background (the
    # splitters follow the sample, not the key range), depths that cross several powers of four (the fourth LSD pass is not skipped).
    import ctypes
    from diff_triangle_rasterization_2D import _C
    _C._lib.ts2d_lab_depth_split.argtypes = [ctypes.c_int, ctypes.c_int]

    def bend(s, kind):
        v = s["vertex"]
        P = v.shape[0]
        zc = v[:, :, 2].mean(axis=1, keepdims=True)
        if kind == "culled":
            v[::3, :, 2] += 5000.0
        elif kind == "one_depth":
            v[: P // 2, :, 2] = zc[: P // 2].mean()  # every vertex of these triangles at ONE z: one depth key, bit for bit
        elif kind == "quantised":
            q = np.round(zc / 20.0) * 20.0
            v[:, :, 2] = q
        elif kind == "background":
            v[: P // 50, :, 2] -= 30000.0  # 2 % of the triangles five octaves behind the rest: the key RANGE is mostly empty, the sample is not
        elif kind == "octaves":
            scale = np.exp2(np.random.default_rng(1).uniform(-3.0, 0.0, size=(P, 1, 1))).astype(np.float32)
            v[:, :, 2] = (s["campos"][2] + (v[:, :, 2] - s["campos"][2]) * scale[:, :, 0])
        return s

    cases = [(12289, "plain"), (20000, "culled"), (93000, "plain"), (93000, "quantised"), (150000, "one_depth"), (300000, "background"), (300000, "octaves"),
             (1000003, "plain"), (1600000, "quantised")]
    for P, kind in cases:
        s = bend(synthetic.scene(P, 320, 200, 1, seed=700 + P % 97), kind)
        got = {}
        for form, (mode, cap) in {"split": (2, 0), "lsd": (1, 0), "split_cap512": (2, 512)}.items():  # 2: the form at every size it supports (the product: to 500 000)
            if form == "split_cap512" and P > 400000:
                continue  # the global-memory path on a million pairs is slow by design; the smaller scenes cover it
            _C._lib.ts2d_lab_depth_split(mode, cap)
            hf = helpers.hip_forward_backward(s, True, backward=False)
            got[form] = (int(hf["num_rendered"]), helpers.hip_state(hf, s, "depth_perm").copy(), helpers.hip_state(hf, s, "point_offsets").copy(),
                         helpers.hip_state(hf, s, "vals").copy() if hf["num_rendered"] > 0 else np.zeros(0, np.int32), hf["out_feature"].copy())
        _C._lib.ts2d_lab_depth_split(0, 0)
        a = got["lsd"]
        e = {"P": P, "kind": kind, "num_rendered": a[0]}
        for form in got:
            if form == "lsd":
                continue
            b = got[form]
            e[form] = float(a[0] != b[0] or not np.array_equal(a[1], b[1]) or not np.array_equal(a[2], b[2]) or not np.array_equal(a[3], b[3])
                            or not np.array_equal(a[4], b[4]))
        out.append(e)
    print("LAB_RESULT " + json.dumps(out))
    sys.exit(0)
if os.environ.get("LAB_SIDE_STREAM") == "1":
    # Round 6's side-stream experiment (lab library only; csrc/api.hip: SideLane): the per-triangle kernel without the SH colours + a colour kernel
    # on a library-owned stream beside the ordering chain.  Same scene with it (ts2d_lab_side_stream) and without (the product's one launch):
    # every state array and every forward output must be IDENTICAL (the colour kernel repeats the single launch's contraction-free expressions),
    # the gradients equal up to the order of the atomics.
    import ctypes
    import torch
    from diff_triangle_rasterization_2D import _C
    _C._lib.ts2d_lab_side_stream.argtypes = [ctypes.c_int]
    for P, W, H, D, variant, feat in [(140000, 640, 360, 3, 2, False), (131072, 400, 300, 1, 3, False), (150001, 320, 200, 1, 2, False), (135000, 320, 200, 3, 3, False)]:
        s = synthetic.scene(P, W, H, D, seed=600 + D)
        if P == 150001:
            s["vertex"][::3, :, 2] += 5000.0  # a third of the triangles behind the camera: culled, the colour kernel skips their SH rows
        got = {}
        for form, on in (("side", 1), ("single", 0)):
            _C._lib.ts2d_lab_side_stream(on)
            hf = helpers.hip_forward_backward(s, True, use_feature=feat, variant=variant)
            got[form] = hf, {k: helpers.hip_state(hf, s, k).copy() for k in ("records", "clamped", "tiles_touched", "rect", "depth", "depth_perm", "point_offsets", "vals", "ranges", "n_contrib")}
        _C._lib.ts2d_lab_side_stream(0)
        (a, sa), (b, sb) = got["side"], got["single"]
        e = {"P": P, "variant": variant, "int_num_rendered": float(a["num_rendered"] != b["num_rendered"])}
        for k in sa:
            e["int_state_" + k] = float(not np.array_equal(sa[k], sb[k]))
        for k in ("out_feature", "depth", "normal", "radii"):
            e["int_" + k] = float(not np.array_equal(a[k], b[k]))
        for k in ("contrib_sum", "contrib_max", "dL_dvertex", "dL_dcenter2D", "dL_dopacity", "dL_dfeature" if feat else "dL_dshs"):
            e[k] = helpers.rel_l2(a[k], b[k])
        out.append(e)
    print("LAB_RESULT " + json.dumps(out))
    sys.exit(0)
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
