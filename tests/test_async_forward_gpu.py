"""Sync-free forward (include/ts2d.h: ts2d_forward; package switch set_instance_capacity): device-side instance count against a
caller-provided capacity.  With enough capacity every output and gradient equals the synchronous path bit for bit (same kernels,
same order; only atomics' summation order may differ); over capacity nothing is emitted and the status word says so."""
import numpy as np
import pytest

import helpers
import synthetic

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [2, 3])
def test_async_forward_equals_synchronous_and_reports_overflow(variant):
    import torch
    import diff_triangle_rasterization_2D as pkg
    s = synthetic.scene(20000, 320, 200, 2, seed=9)
    ref = helpers.hip_forward_backward(s, True, variant=variant)
    N = ref["num_rendered"]
    try:
        pkg.set_instance_capacity(lambda P, W, H: N + 1000)
        got = helpers.hip_forward_backward(s, True, variant=variant)
        assert got["num_rendered"] == N + 1000  # the capacity: only sizes the state for backward
        for k in ("out_feature", "depth", "normal", "radii"):
            assert np.array_equal(got[k], ref[k]), k
        for k in ("contrib_sum", "contrib_max", "dL_dvertex", "dL_dcenter2D", "dL_dshs", "dL_dopacity"):
            assert helpers.rel_l2(got[k], ref[k]) < 1e-6, k
        # the tile ranges (and with them the instance list they index) are identical
        assert np.array_equal(helpers.hip_state(got, s, "ranges"), helpers.hip_state(ref, s, "ranges"))
        assert np.array_equal(helpers.hip_state(got, s, "vals")[:N], helpers.hip_state(ref, s, "vals"))
        # status of a fitting and of an overflowing forward
        from diff_triangle_rasterization_2D import TriangleRasterizer as R2
        from diff_triangle_rasterization_3D import TriangleRasterizer as R3
        rs = helpers.hip_settings(s, rich_info=True)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        R = R3 if variant == 3 else R2
        args = (t(s["vertex"]).requires_grad_(True), torch.zeros((20000, 2), device="cuda", requires_grad=True), t(s["opacity"]))
        out = R(rs)(*args, shs=t(s["shs"]))
        assert pkg.forward_overflowed(out[0]) == (False, N)
        pkg.set_instance_capacity(N - 1)
        out = R(rs)(*args, shs=t(s["shs"]))
        over, n_true = pkg.forward_overflowed(out[0])
        assert over and n_true == N
        bg = t(s["background"])[:, None, None].expand_as(out[0])
        assert torch.equal(out[0], bg) and float(out[4].abs().sum()) == 0.0  # background only, no contributions
        out[0].sum().backward()  # the backward of an overflowed forward is well defined: zero gradients
        assert float(args[0].grad.abs().sum()) == 0.0
    finally:
        pkg.set_instance_capacity(None)


def test_concurrent_forwards_read_their_own_instance_count():
    """The instance count reaches the host through a ring of pinned words (api.hip, acquire_early_count): forwards running at the
    same time on different streams and threads must each get their own."""
    import threading
    import torch
    scenes = [synthetic.scene(20_000 + 7_000 * i, 320, 200, 1, seed=50 + i) for i in range(3)]
    want = [helpers.hip_forward_backward(s, True, False, backward=False)["num_rendered"] for s in scenes]
    assert len(set(want)) == 3
    errors = []

    def worker(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(25):
                    got = helpers.hip_forward_backward(scenes[i], True, False, backward=False)["num_rendered"]
                    if got != want[i]:
                        errors.append((i, got, want[i]))
            st.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


@pytest.mark.parametrize("variant", [2, 3])
def test_a_training_step_replays_from_one_hip_graph(variant):
    """The sync-free forward has no host read, so a whole step -- forward, the loss's upstream gradients, backward -- can be captured ONCE into
    a HIP graph (torch.cuda.CUDAGraph over the library's launches and torch's allocations) and replayed: one graph launch per step instead of
    ~20 kernel launches, ~15 allocations and the autograd engine (bench.py --hip-graph; what bounds a 10 k-triangle step is the host).  A
    replay on NEW parameter values (written in place, as an optimizer does) must equal the eager step on those values."""
    import torch
    import diff_triangle_rasterization_2D as pkg
    from diff_triangle_rasterization_2D import TriangleRasterizer as R2
    from diff_triangle_rasterization_3D import TriangleRasterizer as R3
    P = 12000
    a = synthetic.scene(P, 256, 192, 2, seed=61)
    b = synthetic.scene(P, 256, 192, 2, seed=62)  # same camera and sizes, other triangles
    want = helpers.hip_forward_backward(b, True, variant=variant)
    cap = 2 * max(want["num_rendered"], helpers.hip_forward_backward(a, True, variant=variant, backward=False)["num_rendered"]) + 4096
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    vertex, shs, opacity = t(a["vertex"]).requires_grad_(True), t(a["shs"]).requires_grad_(True), t(a["opacity"]).requires_grad_(True)
    gi, gd, gn = t(a["dL_dout_feature"]), t(a["dL_dout_depth"]), t(a["dL_dout_normal"])
    raster = (R3 if variant == 3 else R2)(helpers.hip_settings(a, True))
    keep = {}

    def step():
        keep.clear()  # the previous step's autograd graph (and its AccumulateGrad nodes) must not outlive it into the capture
        c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
        out = raster(vertex, c2d, opacity, shs=shs)
        torch.autograd.backward([out[0], out[2], out[3]], [gi, gd, gn])
        keep.update(out=out, c2d=c2d)

    try:
        pkg.set_instance_capacity(cap)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                vertex.grad = shs.grad = opacity.grad = None
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        vertex.grad = shs.grad = opacity.grad = None
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        with torch.no_grad():  # an optimizer's in-place update
            vertex.copy_(t(b["vertex"])); shs.copy_(t(b["shs"])); opacity.copy_(t(b["opacity"]))
            gi.copy_(t(b["dL_dout_feature"])); gd.copy_(t(b["dL_dout_depth"])); gn.copy_(t(b["dL_dout_normal"]))
        graph.replay()
        torch.cuda.synchronize()
        out = keep["out"]
        assert np.array_equal(out[1].cpu().numpy(), want["radii"])
        for k, got in (("out_feature", out[0]), ("depth", out[2]), ("normal", out[3])):
            assert np.array_equal(got.detach().cpu().numpy(), want[k]), k
        for k, got in (("contrib_sum", out[4]), ("contrib_max", out[5]), ("dL_dvertex", vertex.grad), ("dL_dshs", shs.grad),
                       ("dL_dopacity", opacity.grad), ("dL_dcenter2D", keep["c2d"].grad)):
            assert helpers.rel_l2(got.detach().cpu().numpy().reshape(want[k].shape), want[k]) < 1e-6, k
        assert pkg.forward_overflowed() == (False, want["num_rendered"])
    finally:
        pkg.set_instance_capacity(None)


def test_graphed_step_trains_like_the_eager_loop():
    """diff_recon_hip.GraphedStep around (render -> loss gradients -> backward) with FusedAdam.step() eager behind every replay (its learning rates and
    bias corrections are host scalars, rewritten every iteration -- VanillaTS_model.py:583 -- so it stays outside the graph): five iterations, two
    cameras alternating through copy_ into the settings' tensors, leave the parameters where the eager loop leaves them; an instance capacity that
    is too small is reported, not silently rendered."""
    import torch
    from diff_recon_hip import FusedAdam, GraphedStep
    from diff_triangle_rasterization_2D import TriangleRasterizer
    P = 9000
    s = synthetic.scene(P, 224, 160, 1, seed=71)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    cams = [synthetic.camera(224, 160), synthetic.camera(224, 160)]
    cams[1]["viewmatrix"] = cams[1]["viewmatrix"].copy()
    cams[1]["viewmatrix"][3, 0] += 1.5  # a second camera: shifted sideways
    cams[1]["projmatrix"] = (cams[1]["viewmatrix"] @ synthetic.projection_matrix(cams[1]["tanfovx"], cams[1]["tanfovy"]).T).astype(np.float32)
    gi, gd, gn = t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])

    def run(graphed, capacity=None):
        rs = helpers.hip_settings(s, True)
        params = {k: t(s[k]).requires_grad_(True) for k in ("vertex", "opacity", "shs")}
        opt = FusedAdam([{"params": [params["vertex"]], "lr": 2e-3}, {"params": [params["opacity"]], "lr": 1e-2}, {"params": [params["shs"]], "lr": 5e-3}],
                        lr=0.0, eps=1e-15)
        seen = {}

        def fwd_bwd():
            c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
            out = TriangleRasterizer(rs)(params["vertex"], c2d, params["opacity"], shs=params["shs"])
            torch.autograd.backward([out[0], out[2], out[3]], [gi, gd, gn])
            seen["n"] = out[0].grad_fn.num_rendered

        def set_camera(c):
            with torch.no_grad():
                rs.viewmatrix.copy_(t(c["viewmatrix"])); rs.projmatrix.copy_(t(c["projmatrix"])); rs.campos.copy_(t(c["campos"]))

        step = None
        if graphed:
            set_camera(cams[0])
            step = GraphedStep(lambda: (opt.zero_grad(set_to_none=True), fwd_bwd())[1], instance_capacity=capacity)
        for it in range(5):
            set_camera(cams[it % 2])
            if graphed:
                step.replay()  # overwrites the (static) .grad tensors the capture left behind
            else:
                opt.zero_grad(set_to_none=True)
                fwd_bwd()
            opt.step()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in params.items()}, seen.get("n"), (step.overflowed() if graphed else None)

    want, n_eager, _ = run(False)
    got, _, status = run(True, capacity=2 * n_eager + 4096)
    assert not status[0] and abs(status[1] - n_eager) <= 0.5 * n_eager  # the true count of the last replay, read back on demand
    for k in want:
        scale = float(want[k].abs().max())
        assert float((got[k] - want[k]).abs().max()) <= 2e-4 * scale, k  # same kernels; the backward's atomic adds are the only freedom
        assert not torch.equal(got[k], t(s[k])), k                       # ... and it trained
    _, _, status = run(True, capacity=max(n_eager // 4, 1))
    assert status[0]  # too small a capacity: reported
