"""Round 6: the forward forks a library-owned side stream (csrc/api.hip: SideLane) for scenes of TS2D_SIDE_STREAM_MIN_TRIANGLES triangles and more --
the render records with their SH colours and the clear of the gradient records run BESIDE the depth sort instead of in front of it -- and the
gradient records live in the geometry state (TS2D_FLAG_PREPARE_BACKWARD / TS2D_FLAG_GRAD_RECORDS_READY).  What must not change: any result."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers
import synthetic

pytestmark = pytest.mark.gpu
LAB_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libts2d_lab.so")
P_BIG = 140_000  # >= TS2D_SIDE_STREAM_MIN_TRIANGLES (131 072)


def test_side_stream_and_single_launch_leave_identical_state_and_outputs():
    if not os.path.exists(LAB_LIB):
        pytest.skip("tools/bin/libts2d_lab.so not built")
    e = dict(os.environ, TS2D_LIBRARY_PATH=LAB_LIB, LAB_SIDE_STREAM="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab_worker.py")], env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("LAB_RESULT ")][-1][len("LAB_RESULT "):])
    assert len(res) == 3
    for case in res:
        for k, v in case.items():
            if k.startswith("int_"):
                assert v == 0.0, (case["P"], case["variant"], k)
            elif k not in ("P", "variant"):
                assert v < 2e-5, (case["P"], case["variant"], k, v)  # fp32 summation order of the atomics only


def _scene_tensors(s, dev="cuda"):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (t(s["vertex"]).requires_grad_(True), t(s["shs"]).requires_grad_(True), t(s["opacity"]).requires_grad_(True),
            [t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])])


@pytest.mark.parametrize("P", [3000, P_BIG])
def test_prepared_gradient_records_equal_the_scratch_path_and_a_second_backward_still_works(P):
    """The forward clears the gradient records inside the geometry state (on the side stream for the large scene, on the caller's stream for the
    small one); the FIRST backward accumulates there and clears nothing, a SECOND backward through the same forward (retain_graph) finds the
    records used and takes the reference's sequence (its own zeroed scratch).  Both must give the gradients of the C ABI's plain form."""
    from diff_triangle_rasterization_2D import TriangleRasterizer, _C, center2D_sink
    s = synthetic.scene(P, 320, 240, 2, seed=21)
    vertex, shs, opacity, g = _scene_tensors(s)
    rs = helpers.hip_settings(s)
    grads = []
    c2d = center2D_sink(P, "cuda")
    out = TriangleRasterizer(rs)(vertex, c2d, opacity, shs=shs)
    node = out[0].grad_fn
    assert node.records_ready
    for _ in range(2):
        torch.autograd.backward([out[0], out[2], out[3]], g, retain_graph=True)
        grads.append([x.grad.clone() for x in (vertex, shs, opacity, c2d)])
        for x in (vertex, shs, opacity, c2d):
            x.grad = None
        assert not node.records_ready
    # the plain form: forward without TS2D_FLAG_PREPARE_BACKWARD, backward on a scratch buffer
    with torch.no_grad():
        args = (rs.image_width, rs.image_height, rs.tanfovx, rs.tanfovy, rs.viewmatrix, rs.projmatrix, rs.campos, rs.sh_degree, rs.gamma, rs.scale_modifier,
                float(rs.background_depth), rs.background, vertex.detach(), shs.detach(), torch.Tensor([]), opacity.detach(), False, True, False)
        n, img, radii, depth, normal, csum, cmax, gb, bb, ib = _C.rasterize_triangles(*args)
        want = _C.rasterize_triangles_backward(*args[2:16], n, radii, gb, bb, ib, *g, True, False)
    assert torch.equal(img, out[0])
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for got in grads:
        assert rel(got[0], want[0]) < 2e-5 and rel(got[1], want[2]) < 2e-5 and rel(got[2], want[4]) < 2e-5 and rel(got[3], want[1]) < 2e-5


def test_center2D_sink_is_a_fresh_leaf_over_cached_zeros():
    from diff_triangle_rasterization_2D import center2D_sink
    a, b = center2D_sink(1000, "cuda"), center2D_sink(1000, "cuda")
    assert a.is_leaf and b.is_leaf and a.requires_grad and a is not b and a.grad is None
    assert a.data_ptr() == b.data_ptr() and float(a.abs().sum()) == 0.0 and a.shape == (1000, 2)
    a.backward(torch.ones_like(a))
    assert b.grad is None and float(a.grad.sum()) == 2000.0


def test_two_forwards_in_flight_on_two_caller_streams_take_different_lanes():
    """Two large forwards + backwards queued on two torch streams without a synchronisation in between: each call forks and joins its own lane."""
    from diff_triangle_rasterization_2D import TriangleRasterizer, center2D_sink
    scenes = [synthetic.scene(P_BIG, 480, 270, 1, seed=31 + i) for i in range(2)]
    ref, res = [], []
    for s in scenes:
        hf = helpers.hip_forward_backward(s, True)
        ref.append(hf)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    keep = []
    for s, st in zip(scenes, streams):
        with torch.cuda.stream(st):
            vertex, shs, opacity, g = _scene_tensors(s)
            c2d = center2D_sink(P_BIG, "cuda")
            out = TriangleRasterizer(helpers.hip_settings(s))(vertex, c2d, opacity, shs=shs)
            torch.autograd.backward([out[0], out[2], out[3]], g)
            keep.append((out, vertex, shs, opacity))
    torch.cuda.synchronize()
    for (out, vertex, shs, opacity), hf in zip(keep, ref):
        assert np.array_equal(out[0].detach().cpu().numpy(), hf["out_feature"]) and np.array_equal(out[1].cpu().numpy(), hf["radii"])
        assert helpers.rel_l2(vertex.grad.cpu().numpy(), hf["dL_dvertex"]) < 2e-5 and helpers.rel_l2(shs.grad.cpu().numpy(), hf["dL_dshs"]) < 2e-5


def test_a_large_step_replays_from_one_hip_graph_with_the_side_stream_inside():
    """The fork / join events pull the library's side stream into a capture of the caller's stream (diff_recon_hip.GraphedStep): replays on new
    parameter values equal the eager step."""
    import diff_triangle_rasterization_2D as pkg
    from diff_recon_hip import GraphedStep
    from diff_triangle_rasterization_2D import TriangleRasterizer, center2D_sink
    s = synthetic.scene(P_BIG, 480, 270, 2, seed=41)
    vertex, shs, opacity, g = _scene_tensors(s)
    rs = helpers.hip_settings(s)
    raster = TriangleRasterizer(rs)
    outs = {}

    def step():
        vertex.grad = shs.grad = opacity.grad = None  # autograd ASSIGNS the gradients: inside the capture they land in the graph's pool
        c2d = center2D_sink(P_BIG, "cuda")
        out = raster(vertex, c2d, opacity, shs=shs)
        torch.autograd.backward([out[0], out[2], out[3]], g)
        outs["img"] = out[0]

    step()
    torch.cuda.synchronize()
    n = int(outs["img"].grad_fn.num_rendered)
    try:
        gs = GraphedStep(step, instance_capacity=int(1.3 * n) + 1024)
        with torch.no_grad():
            opacity.mul_(0.9)
            vertex.add_(0.01)
        gs.replay()
        torch.cuda.synchronize()
        assert not gs.overflowed()[0]
        got = [x.grad.clone() for x in (vertex, shs, opacity)]
        img = outs["img"].detach().clone()
    finally:
        pkg.set_instance_capacity(None)
    step()
    torch.cuda.synchronize()
    assert torch.equal(img, outs["img"].detach())
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    for a, b in zip(got, (vertex.grad, shs.grad, opacity.grad)):
        assert rel(a, b) < 2e-5
