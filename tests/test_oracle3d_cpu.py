"""CPU tests of the oracle's 3D variant (rasterizer_type "3D"; oracle/ts2d_oracle.c, R3D citations there).

Parity status: like the 2D path, the 3D restatement is pinned on the GPU against the reference's own 3D extension built
for gfx950 (tests/test_reference_gpu.py); on the CPU through (a) the shared SH / camera fixtures of test_oracle_cpu.py and
(b) the independent float64 autograd model below."""
import numpy as np

import helpers
import ref3d_f64
import synthetic


def _scene(P=40, W=48, H=40, D=1, seed=31):
    s = synthetic.scene(P, W, H, D, seed=seed, edge_px=9.0)
    # opacity == 1: alpha = min(0.99, G), so the backward's "G < 1/255" skip (R3D backward.cu:351) coincides with the
    # forward's "alpha < 1/255" (forward.cu:265) and the hand-written backward is the derivative of the forward.
    s["opacity"] = np.ones_like(s["opacity"])
    s["shs"] = (s["shs"] * 0.8).astype(np.float32)
    s["background"] = np.array([0.1, 0.3, 0.2], np.float32)
    return s


def test_3d_forward_and_backward_match_float64_autograd():
    P, W, H, D = 40, 48, 40, 1
    s = _scene(P, W, H, D)
    of = helpers.oracle_forward(s, variant=3)
    ob = helpers.oracle_backward(s, of)
    st = of["state"]
    assert (of["radii"] > 0).sum() >= P - 2
    order = np.argsort(st.field("depth"), kind="stable")
    pairs = ref3d_f64.processed_pairs(st, P, W, H)
    img, dep, nor, gv, gsh, gop = ref3d_f64.loss_and_grads(s, D, order, pairs)
    np.testing.assert_allclose(img, of["out_feature"], atol=5e-5)
    np.testing.assert_allclose(dep, of["depth"], rtol=5e-5, atol=1e-3)
    assert helpers.rel_l2(of["normal"], nor) < 1e-4
    assert helpers.rel_l2(ob["dL_dshs"], gsh) < 1e-4
    assert helpers.rel_l2(ob["dL_dvertex"], gv) < 2e-3
    # dL_dopacity is deliberately NOT the derivative where alpha is clamped at 0.99 (R3D backward.cu:451 is ungated)


def test_3d_state_conventions():
    """Unnormalised normal, centroid depth key, 3x-dilated bounding rectangle (R3D forward.cu:93-145)."""
    s = _scene(60, 64, 48, 0, seed=3)
    of = helpers.oracle_forward(s, variant=3)
    st = of["state"]
    vis = of["radii"] > 0
    view = s["viewmatrix"].astype(np.float64)
    vv = (np.concatenate([s["vertex"], np.ones_like(s["vertex"][..., :1])], -1).astype(np.float64) @ view)[..., :3]
    n = np.cross(vv[:, 1] - vv[:, 0], vv[:, 2] - vv[:, 0])
    np.testing.assert_allclose(st.field("normal_view")[vis], n[vis], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(st.field("depth")[vis], vv.mean(1)[vis, 2], rtol=1e-5)
    np.testing.assert_allclose(st.field("v2_view")[vis], vv[vis, 1], rtol=1e-5, atol=1e-6)
    # 2D and 3D variants bin differently (3x dilation of the world-space triangle vs the 2D dilated screen triangle)
    of2 = helpers.oracle_forward(s, variant=2)
    assert of["num_rendered"] > 0 and of2["num_rendered"] > 0


def test_3d_termination_and_quirk_paths_are_exercised():
    """Dense scene: early termination, the 0.99 clamp and G >= 1/255 > alpha pairs all occur; forward/backward finite."""
    s = synthetic.scene(3000, 96, 96, 1, seed=8, edge_px=6.0)
    of = helpers.oracle_forward(s, variant=3)
    ob = helpers.oracle_backward(s, of)
    assert of["state"].field("final_T").min() <= 1e-4
    for k, v in ob.items():
        assert np.isfinite(v).all(), k
    assert np.isfinite(of["out_feature"]).all() and np.isfinite(of["depth"]).all()
