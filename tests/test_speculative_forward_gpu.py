"""The package's default forward (include/ts2d.h: ts2d_forward_speculative): everything is queued for a GUESSED instance capacity, the
exact num_rendered comes back to the host behind the queue.  Whatever the guess was -- none (first call), right, far too small -- the
results are those of the reference's sequence: num_rendered exact, integer state and images bit-identical, gradients equal up to the
atomics' summation order."""
import ctypes as C

import numpy as np
import pytest

import helpers
import synthetic

pytestmark = pytest.mark.gpu

W, H = 211, 173  # an image size no other test uses: the capacity history is keyed by (device, variant, width, height)


def _hint(P, variant):
    from diff_triangle_rasterization_2D import _C
    return int(_C._lib.ts2d_instance_capacity_hint(P, W, H, 16 if variant == 3 else 0))


def _same(a, b):
    for k in ("out_feature", "depth", "normal", "radii"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("contrib_sum", "contrib_max", "dL_dvertex", "dL_dcenter2D", "dL_dshs", "dL_dopacity"):
        assert helpers.rel_l2(a[k], b[k]) < 1e-6, k


@pytest.mark.parametrize("variant", [2, 3])
def test_guess_none_right_and_too_small_give_the_same_results(variant):
    P = 9000
    small = synthetic.scene(P, W, H, 2, seed=3, edge_px=3.0)
    large = synthetic.scene(P, W, H, 2, seed=4, edge_px=14.0)
    o_small = helpers.oracle_forward(small, True, variant=variant)
    o_large = helpers.oracle_forward(large, True, variant=variant)
    assert o_large["num_rendered"] > 2 * o_small["num_rendered"]  # 1.25 x the small scene's count cannot hold the large one

    first = helpers.hip_forward_backward(small, True, variant=variant)     # no history for this size: exact second half
    assert first["num_rendered"] == o_small["num_rendered"]
    h = _hint(P, variant)
    assert o_small["num_rendered"] < h <= 1.25 * o_small["num_rendered"] * 1.07 + 8192  # 1.25 x + slack, quantised upwards by < 1/16
    again = helpers.hip_forward_backward(small, True, variant=variant)     # guess fits: the speculative launches ARE the forward
    assert again["num_rendered"] == o_small["num_rendered"]
    _same(again, first)
    assert again["buffers"][1].numel() > first["buffers"][1].numel()       # the binning buffer was sized for the capacity, not the count
    for name in ("ranges", "vals", "keys", "n_contrib"):
        assert np.array_equal(helpers.hip_state(again, small, name), helpers.hip_state(first, small, name)), name

    over = helpers.hip_forward_backward(large, True, variant=variant)      # guess too small: nothing emitted, exact re-run of the second half
    assert over["num_rendered"] == o_large["num_rendered"]
    assert np.array_equal(over["radii"], o_large["radii"])
    assert helpers.rel_l2(over["out_feature"], o_large["out_feature"]) < 1e-4
    st = o_large["state"]
    assert np.array_equal(helpers.hip_state(over, large, "vals").astype(np.int64).reshape(-1), st.field("vals").astype(np.int64).reshape(-1))
    fits = helpers.hip_forward_backward(large, True, variant=variant)      # the history now knows the large count
    _same(fits, over)
    # the decaying maximum keeps the large scene's capacity for the views that follow it
    assert _hint(P, variant) >= o_large["num_rendered"]
    back = helpers.hip_forward_backward(small, True, variant=variant)
    _same(back, first)


def test_speculative_c_abi_contract():
    """ts2d_forward_speculative straight through the C ABI: with no binning buffer it is ts2d_forward_bin (the count comes back, nothing is
    rendered); with a buffer that is too small *num_rendered exceeds ts2d_binning_capacity and the image is the background."""
    import torch
    from diff_triangle_rasterization_2D import _C
    L = _C._lib
    s = synthetic.scene(5000, W, H, 1, seed=12)
    want = helpers.oracle_forward(s, True)["num_rendered"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    view, proj, campos, bg = t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), t(s["background"])
    vertex, shs, opacity = t(s["vertex"]), t(s["shs"]), t(s["opacity"])
    P, M = 5000, shs.shape[1]
    cam = _C._Camera(W, H, s["tanfovx"], s["tanfovy"], view.data_ptr(), proj.data_ptr(), campos.data_ptr())
    geom = _C._Geometry(P, 1, M, 3, 1.0, 1.0, 5000.0, bg.data_ptr(), vertex.data_ptr(), shs.data_ptr(), None, opacity.data_ptr())
    flags = _C.FLAG_RICH_INFO | _C.FLAG_USE_SHS
    u8 = dict(device="cuda", dtype=torch.uint8)
    g = torch.empty(L.ts2d_geometry_state_bytes(P), **u8)
    im = torch.empty(L.ts2d_image_state_bytes(W, H), **u8)
    img, depth, normal = torch.full((3, H, W), 7.0, device="cuda"), torch.empty((H, W), device="cuda"), torch.empty((3, H, W), device="cuda")
    csum, cmax, radii = torch.empty(P, device="cuda"), torch.empty(P, device="cuda"), torch.empty(P, device="cuda", dtype=torch.int32)
    out = _C._ForwardOut(img.data_ptr(), depth.data_ptr(), normal.data_ptr(), csum.data_ptr(), cmax.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    n = C.c_int64(-1)
    st = _C._State(g.data_ptr(), g.numel(), None, 0, im.data_ptr(), im.numel())
    assert L.ts2d_forward_speculative(C.byref(cam), C.byref(geom), flags, radii.data_ptr(), C.byref(st), C.byref(out), C.byref(n), stream) == 0
    torch.cuda.synchronize()
    assert n.value == want and float(img.min()) == 7.0  # counted, not rendered
    small = torch.empty(L.ts2d_binning_state_bytes(want // 2, W, H), **u8)
    assert L.ts2d_binning_capacity(small.numel(), W, H) < want
    st = _C._State(g.data_ptr(), g.numel(), small.data_ptr(), small.numel(), im.data_ptr(), im.numel())
    assert L.ts2d_forward_speculative(C.byref(cam), C.byref(geom), flags, radii.data_ptr(), C.byref(st), C.byref(out), C.byref(n), stream) == 0
    torch.cuda.synchronize()
    assert n.value == want
    assert torch.equal(img, bg[:, None, None].expand_as(img)) and float(csum.abs().sum()) == 0.0  # overflow: background, no contributions
    over, n_true = C.c_int32(0), C.c_int64(0)
    assert L.ts2d_forward_status(C.byref(st), P, W, H, C.byref(over), C.byref(n_true), stream) == 0 and over.value == 1 and n_true.value == want
    exact = torch.empty(L.ts2d_binning_state_bytes(want, W, H), **u8)
    st = _C._State(g.data_ptr(), g.numel(), exact.data_ptr(), exact.numel(), im.data_ptr(), im.numel())
    assert L.ts2d_forward_render(C.byref(cam), C.byref(geom), flags, want, C.byref(st), C.byref(out), stream) == 0
    assert L.ts2d_forward_status(C.byref(st), P, W, H, C.byref(over), C.byref(n_true), stream) == 0 and over.value == 0
    ref = helpers.oracle_forward(s, True)
    assert helpers.rel_l2(img.cpu().numpy(), ref["out_feature"]) < 1e-4
    # a non-NULL binning buffer too small for any instance is refused (it would queue no render, and a scene of zero instances
    # would then pass the caller's  num_rendered > capacity  test with 0 > 0 and leave the outputs unwritten)
    tiny = torch.empty(64, **u8)
    assert L.ts2d_binning_capacity(tiny.numel(), W, H) == 0
    st = _C._State(g.data_ptr(), g.numel(), tiny.data_ptr(), tiny.numel(), im.data_ptr(), im.numel())
    assert L.ts2d_forward_speculative(C.byref(cam), C.byref(geom), flags, radii.data_ptr(), C.byref(st), C.byref(out), C.byref(n), stream) == 3  # TS2D_ERR_CAPACITY
    assert b"too small for any instance" in L.ts2d_last_error()


def test_capacity_hint_keys_and_the_overflow_counter():
    """the histories behind ts2d_instance_capacity_hint are kept per caller key, so a stream of small views (key 1) is not sized by
    a stream of large ones (key 2) of the same image size; every speculative forward whose guess was too small is counted."""
    import diff_triangle_rasterization_2D as pkg
    P = 7000
    small = synthetic.scene(P, 208, 176, 1, seed=31, edge_px=3.0)
    large = synthetic.scene(P, 208, 176, 1, seed=32, edge_px=14.0)
    hint = lambda: int(pkg._C._lib.ts2d_instance_capacity_hint(P, 208, 176, 0))
    try:
        pkg.set_capacity_hint_key(1)
        n_small = helpers.hip_forward_backward(small, True, backward=False)["num_rendered"]
        pkg.set_capacity_hint_key(2)
        assert hint() == 0                                   # key 2 has no history although key 1 rendered this size
        n_large = helpers.hip_forward_backward(large, True, backward=False)["num_rendered"]
        assert n_large > 2 * n_small and hint() >= n_large
        pkg.set_capacity_hint_key(1)
        assert n_small < hint() < n_large                    # ... and key 1's history is untouched by the large views
        before = pkg.speculative_overflows()
        helpers.hip_forward_backward(small, True, backward=False)   # fits
        assert pkg.speculative_overflows() == before
        got = helpers.hip_forward_backward(large, True, backward=False)  # key 1's guess is too small for the large view: counted, result exact
        assert pkg.speculative_overflows() == before + 1 and got["num_rendered"] == n_large
    finally:
        pkg.set_capacity_hint_key(0)


def test_capture_through_render_view_keeps_the_chain_rule():
    """GradBucket.capture() with NON-LEAF rasterizer inputs (sigmoid(raw_opacity), cat(f_dc, f_rest), rescaled vertices:
    diff_recon_hip.render_view): the parameters upstream still receive their gradients, equal to a run without a capture."""
    import torch
    from diff_recon_hip import render_view
    from diff_triangle_rasterization_2D.parallel import GradBucket
    s = synthetic.scene(4000, 160, 120, 1, seed=21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    class Cam:  # what TriangleRenderer reads from the reference's Camera (src/diff_recon/utils/camera.py)
        image_width, image_height = 160, 120
        tan_fovx, tan_fovy = s["tanfovx"], s["tanfovy"]
        world_view_transform, full_proj_transform, camera_center = t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"])
        device = "cuda"

    def run(captured):
        vertex = t(s["vertex"]).requires_grad_(True)
        f_dc = t(s["shs"][:, :1]).requires_grad_(True)
        f_rest = t(s["shs"][:, 1:]).requires_grad_(True)
        raw = torch.logit(t(s["opacity"]).clamp(1e-3, 1 - 1e-3)).requires_grad_(True)
        bucket = GradBucket([vertex.shape, raw.shape, torch.Size((4000, 2)), torch.Size((4000, 4, 3))], "cuda",
                            names=["vertex", "opacity", "center2D", "color"])
        import contextlib
        with (bucket.capture() if captured else contextlib.nullcontext()):
            pkg = render_view(Cam, vertex, f_dc, f_rest, raw, bg_color=t(s["background"]), gamma=1.0, active_sh_degree=1, max_sh_degree=1,
                              gamma_rescale=True, rasterizer_type="2D")
            (pkg["render"] * t(s["dL_dout_feature"])).sum().backward()
        return [p.grad for p in (vertex, f_dc, f_rest, raw)], bucket

    plain, _ = run(False)
    cap, bucket = run(True)
    for a, b, name in zip(cap, plain, ("vertex", "f_dc", "f_rest", "raw_opacity")):
        assert a is not None, f"{name} lost its gradient under capture()"
        assert float((a - b).norm() / b.norm()) < 1e-5, name
    # the bucket holds the gradient with respect to the rasterizer's inputs (activated opacity, rescaled vertices, concatenated SH)
    nv = bucket.named_views()
    assert float(nv["vertex"].abs().sum()) > 0 and float(nv["opacity"].abs().sum()) > 0 and float(nv["color"].abs().sum()) > 0


@pytest.mark.parametrize("variant", [2, 3])
def test_background_depth_as_a_device_tensor_needs_no_host_value(variant):
    """settings.background_depth may be the 0-dim device tensor the reference's model computes every step (VanillaTS_model.py:623).  The
    reference's binding converts it to a host float (a device synchronisation per forward); here it reaches the kernels as a pointer
    (ts2d_geometry.background_depth_dev).  Same outputs and gradients, bit for bit, as the float."""
    import torch
    if variant == 3:
        from diff_triangle_rasterization_3D import TriangleRasterizer
    else:
        from diff_triangle_rasterization_2D import TriangleRasterizer
    s = synthetic.scene(6000, 160, 96, 1, seed=31)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def run(bg_depth):
        rs = helpers.hip_settings(s, rich_info=True)._replace(background_depth=bg_depth)
        vertex, opacity, shs = t(s["vertex"]).requires_grad_(True), t(s["opacity"]).requires_grad_(True), t(s["shs"]).requires_grad_(True)
        c2d = torch.zeros((6000, 2), device="cuda", requires_grad=True)
        out = TriangleRasterizer(rs)(vertex, c2d, opacity, shs=shs)
        ((out[0] * t(s["dL_dout_feature"])).sum() + (out[2] * t(s["dL_dout_depth"])).sum() + (out[3] * t(s["dL_dout_normal"])).sum()).backward()
        return [out[0].detach(), out[2].detach(), vertex.grad, opacity.grad, shs.grad]

    value = 4321.125
    as_float = run(value)
    as_tensor = run(torch.tensor(value, device="cuda"))            # 0-dim, like (campos - vertex).norm(dim=-1).max()
    as_double = run(torch.tensor(value, device="cuda", dtype=torch.float64))
    assert not torch.equal(run(1000.0)[1], as_float[1])           # the value matters: T * background_depth is part of every depth pixel
    for a, b, c in zip(as_float, as_tensor, as_double):
        assert torch.equal(a, b) or float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())  # gradients: atomics' summation order
        assert torch.equal(a, c) or float((a - c).abs().max()) <= 1e-6 * float(a.abs().max())
    assert torch.equal(as_float[0], as_tensor[0]) and torch.equal(as_float[1], as_tensor[1])


def test_nothing_visible_after_a_history_exists():
    """A guessed capacity > 0 and a true count of 0 (every triangle behind the camera): the speculative launches run over an empty list --
    background image, zero radii / statistics, zero gradients -- and num_rendered is 0."""
    import torch
    w, h = 203, 131  # own image size: own capacity history
    s = synthetic.scene(5000, w, h, 1, seed=5)
    first = helpers.hip_forward_backward(s, True)
    assert first["num_rendered"] > 0
    from diff_triangle_rasterization_2D import _C
    assert int(_C._lib.ts2d_instance_capacity_hint(5000, w, h, 0)) > 0
    hidden = dict(s)
    hidden["vertex"] = s["vertex"].copy()
    hidden["vertex"][:, :, 2] += 5000.0  # behind the camera at z = 1200 looking down -z
    got = helpers.hip_forward_backward(hidden, True)
    assert got["num_rendered"] == 0 and not got["radii"].any()
    assert np.array_equal(got["out_feature"], np.broadcast_to(s["background"][:, None, None], got["out_feature"].shape))
    assert not got["contrib_sum"].any() and not got["normal"].any()
    for k in ("dL_dvertex", "dL_dshs", "dL_dopacity", "dL_dcenter2D"):
        assert not got[k].any(), k
    again = helpers.hip_forward_backward(s, True)  # and the visible scene still renders as before
    assert again["num_rendered"] == first["num_rendered"] and np.array_equal(again["out_feature"], first["out_feature"])
