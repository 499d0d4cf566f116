"""Image-parallel equivalence through the REAL rasterizer: two ranks (two processes sharing the one GPU of the test box, gloo
transport) render one view each, exchange gradients exactly as bench.py --gpus N does (backward kernels writing into the
exchange bucket, factored SH-gradient exchange, reduced render statistics), and the result must equal one process rendering both
views and summing.  Hardware with >= 2 GPUs runs the same code over RCCL (bench.py); this test pins the protocol and the
"2 ranks x 1 view == 1 rank x 2 views" semantics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

P, W, H, D = 6000, 200, 144, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _view(rank):
    import synthetic
    cam = synthetic.camera(W, H)
    if rank > 0:
        view = cam["viewmatrix"].copy()
        shift = np.array([9.0 * rank, -4.0 * rank, 0.0], np.float32)
        view[3, :3] -= shift * np.array([-1, 1, -1], np.float32)
        cam["viewmatrix"] = view
        cam["projmatrix"] = (view @ synthetic.projection_matrix(cam["tanfovx"], cam["tanfovy"]).T).astype(np.float32)
        cam["campos"] = np.array([0, 0, synthetic.CAM_DIST], np.float32) + shift
    return cam


def _render(s, cam, variant, dev, vertex, shs, opacity, bucket=None, sink=None):
    from diff_triangle_rasterization_2D import TriangleRasterizationSettings, parallel
    if variant == 3:
        from diff_triangle_rasterization_3D import TriangleRasterizer
    else:
        from diff_triangle_rasterization_2D import TriangleRasterizer
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = TriangleRasterizationSettings(
        image_width=W, image_height=H, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], viewmatrix=t(cam["viewmatrix"]),
        projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), sh_degree=D, gamma=1.0, scale_modifier=1.0, background_depth=50.0,
        background=t(s["background"]), back_culling=False, rich_info=True, debug=False)
    c2d = torch.zeros((P, 2), device=dev, requires_grad=True)
    g = [t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])]
    if bucket is not None:
        with bucket.capture(), parallel.factored_sh_grads(sink):
            out = TriangleRasterizer(rs)(vertex, c2d, opacity, shs=shs)
            torch.autograd.backward([out[0], out[2], out[3]], g)
    else:
        out = TriangleRasterizer(rs)(vertex, c2d, opacity, shs=shs)
        torch.autograd.backward([out[0], out[2], out[3]], g)
    return out, c2d


def _worker(rank, world, port, variant, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [root, os.path.join(root, "triangle-splatting_amd"), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import synthetic
        from diff_triangle_rasterization_2D import parallel
        dev = torch.device("cuda", 0)
        s = synthetic.scene(P, W, H, D, seed=77)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        M = s["shs"].shape[1]
        mk = lambda: (t(s["vertex"]).requires_grad_(True), t(s["shs"]).requires_grad_(True), t(s["opacity"]).requires_grad_(True))

        # --- this rank's view, exchanged like bench.py --gpus 2 ---
        vertex, shs, opacity = mk()
        bucket = parallel.GradBucket([vertex.shape, opacity.shape, torch.Size((P, 2))], dev, names=["vertex", "opacity", "center2D"])
        sink = parallel.ShGradSink()
        out, _ = _render(s, _view(rank), variant, dev, vertex, shs, opacity, bucket, sink)
        bucket.reduce_async()
        shs_grad = parallel.exchange_factored_sh_grads(sink, vertex, D, M)
        g_vertex, g_opacity, g_c2d = [x.clone() for x in bucket.wait()]
        stats = parallel.reduce_render_stats({"radii": out[1].clone(), "contrib_sum": out[4].clone(), "contrib_max": out[5].clone(),
                                              "visible_count": (out[1] > 0).to(torch.int32)})

        # --- reference: this process renders BOTH views and sums ---
        v2, s2, o2 = mk()
        c2d_sum = torch.zeros((P, 2), device=dev)
        ref_stats = None
        for r in range(world):
            o, c2d = _render(s, _view(r), variant, dev, v2, s2, o2)
            c2d_sum += c2d.grad
            cur = {"radii": o[1], "contrib_sum": o[4], "contrib_max": o[5], "visible_count": (o[1] > 0).to(torch.int32)}
            ref_stats = cur if ref_stats is None else {k: (ref_stats[k] + cur[k] if "count" in k else torch.maximum(ref_stats[k], cur[k]))
                                                       for k in cur}
        rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
        errs = {"vertex": rel(g_vertex, v2.grad), "opacity": rel(g_opacity, o2.grad), "center2D": rel(g_c2d, c2d_sum), "shs": rel(shs_grad, s2.grad)}
        ok = all(e < 2e-5 for e in errs.values())  # fp32 summation order only (atomics, all-reduce)
        ok = ok and all(torch.equal(stats[k], ref_stats[k]) for k in ("radii", "visible_count"))
        ok = ok and all(torch.allclose(stats[k], ref_stats[k], rtol=1e-6, atol=0) for k in ("contrib_sum", "contrib_max"))
        ok = ok and shs.grad is None  # factored: the dense per-view dL_dshs was never formed
        q.put((rank, bool(ok), errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", [2, 3])
def test_two_ranks_one_view_each_equals_one_rank_two_views(variant):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, variant, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get(timeout=10) for _ in range(2)]
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("variant", [2, 3])
def test_two_views_under_one_capture_are_not_double_counted(variant):
    """Several views per rank under ONE capture with LEAF parameters (the reference's triangle_renderer passes model._vertex itself,
    zero_grad(set_to_none=True) leaves .grad None): the bucket must hold g1 + g2 -- not g1 + 2 g2, which is what happens when autograd's
    AccumulateGrad aliases param.grad to the bucket's view --, the parameters' .grad must stay untouched and every view's own center2D
    tensor must get its own gradient (the per-view densification statistic)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "triangle-splatting_amd")]
    import synthetic
    from diff_triangle_rasterization_2D import parallel
    dev = torch.device("cuda", 0)
    s = synthetic.scene(P, W, H, D, seed=78)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: (t(s["vertex"]).requires_grad_(True), t(s["shs"]).requires_grad_(True), t(s["opacity"]).requires_grad_(True))
    vertex, shs, opacity = mk()
    bucket = parallel.GradBucket([vertex.shape, opacity.shape, torch.Size((P, 2)), shs.shape], dev, names=["vertex", "opacity", "center2D", "color"])
    c2ds = []
    with bucket.capture():
        for r in range(2):
            _, c2d = _render(s, _view(r), variant, dev, vertex, shs, opacity)
            c2ds.append(c2d)
    assert vertex.grad is None and opacity.grad is None and shs.grad is None  # the gradients live in the bucket
    got = dict(zip(("vertex", "opacity", "center2D", "color"), bucket.wait()))
    v2, s2, o2 = mk()
    ref_c2d = []
    for r in range(2):
        _, c2d = _render(s, _view(r), variant, dev, v2, s2, o2)
        ref_c2d.append(c2d.grad.clone())
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    assert rel(got["vertex"], v2.grad) < 2e-5 and rel(got["opacity"], o2.grad) < 2e-5 and rel(got["color"], s2.grad) < 2e-5
    assert rel(got["center2D"], ref_c2d[0] + ref_c2d[1]) < 2e-5
    for mine, ref in zip(c2ds, ref_c2d):  # per-view statistic, untouched by the later view
        assert mine.grad is not None and mine.grad.data_ptr() != got["center2D"].data_ptr()
        assert rel(mine.grad, ref) < 2e-5


def _rccl_world1_worker(port, q):
    """The RCCL branch of the exchange on ONE GPU: a process group of one rank over the nccl (= RCCL) backend, collectives forced."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [root, os.path.join(root, "triangle-splatting_amd"), here]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import synthetic
        from diff_triangle_rasterization_2D import parallel
        assert dist.get_backend() == "nccl"
        s = synthetic.scene(P, W, H, D, seed=79)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        M = s["shs"].shape[1]
        vertex, shs, opacity = t(s["vertex"]).requires_grad_(True), t(s["shs"]).requires_grad_(True), t(s["opacity"]).requires_grad_(True)
        # reference: plain backward, no bucket
        _, c2d = _render(s, _view(0), 2, dev, vertex, shs, opacity)
        want = [vertex.grad.clone(), opacity.grad.clone(), c2d.grad.clone(), shs.grad.clone()]
        vertex.grad = opacity.grad = shs.grad = None
        errs = {}
        rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
        for mode in ("rs_ag", "all_reduce"):
            bucket = parallel.GradBucket([vertex.shape, opacity.shape, torch.Size((P, 2))], dev, names=["vertex", "opacity", "center2D"], mode=mode,
                                         force_collectives=True)
            _, sh_group = None, dist.new_group(backend="nccl")  # what parallel.exchange_groups() creates at world > 1
            shx = parallel.FactoredShExchange(sh_group, dev)
            for it in range(3):  # repeated WITHOUT host synchronisation: the side streams must order themselves against the compute stream
                sink = parallel.ShGradSink()
                _render(s, _view(0), 2, dev, vertex, shs, opacity, bucket, sink)
                bucket.reduce_async()           # reduce_scatter_tensor into the rank's own slice of the SAME buffer, all_gather back (side stream)
                shx.start(sink, vertex, D, M, uniform=True)
                got = bucket.wait()
                got_shs = shx.wait()
                got[0].mul_(1.0)                # compute-stream work right behind the wait
            torch.cuda.synchronize()
            errs[mode] = [rel(g, w) for g, w in zip(list(got) + [got_shs], want)]
            assert bucket.flat.numel() % 4 == 0 and vertex.grad is None
            # round 5: the visible-rows exchange (parallel.VisibleRows) on the device -- mask all-reduce on its own side stream right behind the
            # forward, compact bucket / compact colour factors through RCCL, sums scattered back; three steps without host synchronisation
            # other than the row count the exchange itself needs
            rows = parallel.VisibleRows(None, dev)
            for it in range(3):
                sink = parallel.ShGradSink()
                out, _ = _render(s, _view(0), 2, dev, vertex, shs, opacity, bucket, sink)
                rows.begin([out[1]])
                bucket.reduce_async(rows=rows)
                shx.start(sink, vertex, D, M, uniform=True, rows=rows)
                got = bucket.wait()
                got_shs = shx.wait()
                got[0].mul_(1.0)
            torch.cuda.synchronize()
            errs[mode + "+visible_rows"] = [rel(g, w) for g, w in zip(list(got) + [got_shs], want)]
            nvis = int((out[1] > 0).sum())
            assert 0 < nvis <= P and int(rows.index().numel()) == nvis and bucket.last_exchanged_bytes <= bucket.padded * 4
            # round 5: the ranged exchange -- the per-triangle kernel of the backward in 4 launches with an event behind each (ts2d_backward_ranged),
            # the bucket reduced range by range on the side stream as the events fire
            rb = parallel.GradBucket([vertex.shape, opacity.shape, torch.Size((P, 2))], dev, names=["vertex", "opacity", "center2D"], force_collectives=True)
            rb.prepare_ranges(4)
            for it in range(3):
                sink = parallel.ShGradSink()
                _render(s, _view(0), 2, dev, vertex, shs, opacity, rb, sink)
                assert rb._ranges_recorded
                rb.reduce_ranges_async()
                shx.start(sink, vertex, D, M, uniform=True)
                got = rb.wait()
                got_shs = shx.wait()
                got[0].mul_(1.0)
            torch.cuda.synchronize()
            errs[mode + "+ranges"] = [rel(g, w) for g, w in zip(list(got) + [got_shs], want)]
            # a SECOND view under the same capture adds into the bucket behind the first backward's range events -- they no longer say
            # "rows final", so the ranged exchange must fall back to ordering itself behind the compute stream as a whole.  Twice the same view:
            # the bucket must hold exactly twice the single view's gradients (a race would show as a wrong sum or a torn bucket).
            for it in range(3):
                sink = parallel.ShGradSink()
                from diff_triangle_rasterization_2D import TriangleRasterizationSettings  # noqa: F401
                with rb.capture(), parallel.factored_sh_grads(sink):
                    for _v in range(2):
                        _render(s, _view(0), 2, dev, vertex, shs, opacity)
                assert not rb._ranges_recorded  # invalidated by the second view's add_
                rb.reduce_ranges_async()
                got = rb.wait()
                got[0].mul_(1.0)
            torch.cuda.synchronize()
            errs[mode + "+ranges, two views"] = [rel(g, 2 * w) for g, w in zip(list(got), want[:3])]
        ok = all(e < 2e-5 for v in errs.values() for e in v)
        q.put((bool(ok), errs))
    finally:
        dist.destroy_process_group()


def test_rccl_reduce_scatter_all_gather_branch_runs_on_one_gpu():
    """The reduce-scatter + all-gather branch of GradBucket (RCCL only: gloo has no reduce_scatter_tensor and takes the all-reduce
    fallback in every other test) and the side-stream SH exchange on a second communicator, executed over the nccl backend in a group
    of one rank (legal in NCCL / RCCL): in-place slice semantics, side-stream ordering against the backward kernels that fill the
    bucket, repeated steps without host synchronisation."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1_worker, args=(_free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    ok, errs = q.get(timeout=10)
    assert ok, errs


def test_bench_gpus_2_spawns_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` -- the N = 1 command with the number changed, NO torchrun in front -- must run two ranks and say
    so.  Two processes share this box's one GPU over gloo (TS2D_BENCH_BACKEND: a functional run, its timings mean nothing); the line must carry
    n_gpus == 2, the backend, both exchange modes (the comparison leg is the default for N > 1 since round 6) and the bytes a rank puts on the wire."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TS2D_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--triangles", "20000", "--width", "320", "--height", "240",
                        "--steps", "3", "--warmup", "1", "--settle-steps", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["distributed"] == {"world": 2, "backend": "gloo", "requested": "gloo"}
    ex = line["config"]["exchange"]
    assert ex["mode"].startswith("synchronous") and ex["other_mode"]["mode"] == "delayed" and ex["other_mode"]["ms_per_step"] > 0
    wire = ex["wire_bytes_per_rank_and_step"]
    assert wire["bucket_rs_ag"] == ex["bucket_bytes_per_rank_and_step"] and wire["sh_factors_all_gather"] == 20000 * 12  # 2 (N-1)/N = 1, (N-1) = 1
    # and the mismatch: a world that contradicts --gpus is refused before anything is measured
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr
