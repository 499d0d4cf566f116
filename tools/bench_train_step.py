#!/usr/bin/env python
"""One TRAINING ITERATION of the reference's loop at the shape of BASELINE configs, on the HIP path end to end (VERDICT r5 "missing" 2 / item 4):

    render_view at render_up_scale x the resolution -> bilinear down-sample -> L1 + SSIM (+ w_geometry x depth / normal consistency)
    -> backward -> FusedAdam step -> per-iteration densification statistics
    (src/diff_recon/models/VanillaTS_model.py:583-685, src/diff_recon/trainers/VanillaTS_trainer.py:218-239, trainer_utils.py:45-103, 204-257)

Prints ONE JSON line per configuration: ms per iteration eager and with render + loss + backward + statistics replayed from one HIP graph
(diff_recon_hip.GraphedStep; the optimizer step stays eager: its learning rates are host scalars the model rewrites every iteration), the raster
step alone at the same size (forward + backward with fixed upstream gradients = what bench.py times), their difference, and for every kernel of
this library outside the rasterizer its average duration, its algorithmic bytes and the fraction of the 8 TB/s HBM roofline that makes.

    python tools/bench_train_step.py --config headline|mesh93k|mesh93k_g50|lego300k [--iters 40]
Algorithmic bytes (f32 elements x 4): photometric_fwd 5 n (2 images in, 3 derivative maps out), photometric_bwd 6 n (5 in, 1 out), n = C h w;
downsample_fwd / _bwd (s^2 + 1) per OUTPUT element; adam_step 28 per parameter float (param, grad, 2 moments in; param, 2 moments out);
depth_normal_fwd 16 per pixel (depth + normal in -- a lower bound: the kernels keep low-resolution intermediates), depth_normal_bwd 32 per pixel
(depth + normal in, their gradients out); training_statistic 68 per triangle (radii, center2D gradient, two contributions in; six running
statistics read and written).
"""
import argparse
import gc
import json
import os
import sys
import time
from types import SimpleNamespace as NS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]
import numpy as np
import torch

import synthetic
import diff_recon_hip as D
from diff_triangle_rasterization_2D import _C
import diff_triangle_rasterization_2D as pkg2d
from diff_triangle_rasterization_2D.parallel import ShGradSink, factored_sh_grads

CONFIGS = {
    # name: triangles, camera width, height, render_up_scale, SH degree stored / active, rasterizer, gamma, w_geometry
    "headline": dict(P=1_000_000, w=1920, h=1080, up=1, D=3, rast="2D", gamma=1.0, w_geo=0.0),      # bench.py's scene as a training iteration (MipNerf360_VanillaTS-like: 2D, SH 3)
    "lego300k": dict(P=300_000, w=800, h=800, up=1, D=3, rast="2D", gamma=1.0, w_geo=0.0),            # BASELINE configs[1]
    "mesh93k": dict(P=93_000, w=800, h=800, up=2, D=0, rast="3D", gamma=1.0, w_geo=0.05),             # BASELINE configs[3]: NerfSynthetic_VanillaTS_mesh (3D, SH 0, render_up_scale 2, geometry loss)
    "mesh93k_g50": dict(P=93_000, w=800, h=800, up=2, D=0, rast="3D", gamma=50.0, w_geo=0.05),        # the same at the end of the gamma schedule
}
HBM = 8000.0


class Camera:
    def __init__(self, s, w, h, dev):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.image_width, self.image_height = w, h
        self.tan_fovx, self.tan_fovy = s["tanfovx"], s["tanfovy"]
        self.world_view_transform, self.full_proj_transform = t(s["viewmatrix"]), t(s["projmatrix"])
        self.camera_center = t(s["campos"])
        self.device = dev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="headline", choices=sorted(CONFIGS))
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--dense-adam", action="store_true", help="the colour parameters stepped from the dense dL_dshs (the form of rounds 4-5) instead of "
                    "the factored gradient (FusedAdam.step(sh_factors=...), include/ts_optim.h: tso_adam_step_sh_factored)")
    ap.add_argument("--mimic-dense-alloc", action="store_true", help="experiment: the factored iteration with the dense iteration's allocation pattern "
                    "(a (P, M, 3) tensor allocated and dropped where the backward would have allocated dL_dshs)")
    ap.add_argument("--hold-mb", type=int, default=0, help="experiment: a tensor of this many MB allocated before anything else and kept")
    a = ap.parse_args()
    c = NS(**CONFIGS[a.config])
    dev = torch.device("cuda")
    W, H = c.w * c.up, c.h * c.up  # the rasterizer's resolution
    hold = torch.empty(a.hold_mb << 20, dtype=torch.uint8, device=dev) if a.hold_mb else None  # noqa: F841
    s = synthetic.scene(c.P, W, H, c.D, seed=42)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    cam = Camera(s, c.w, c.h, dev)
    M = (c.D + 1) ** 2
    vertex = torch.nn.Parameter(t(s["vertex"]))
    shs = torch.nn.Parameter(t(s["shs"]))  # ONE (P, M, 3) colour tensor (FusedAdam lr / lr_tail): no torch.cat per forward
    raw_opacity = torch.nn.Parameter(torch.logit(t(s["opacity"]).clamp(0.02, 0.98)))
    groups = [{"params": [vertex], "lr": 0.0, "name": "vertex"}, {"params": [raw_opacity], "lr": 0.0, "name": "opacity"},
              {"params": [shs], "lr": 0.0, "lr_tail": 0.0, "tail_period": 3 * M, "tail_split": 3, "name": "shs"}]
    opt = D.FusedAdam(groups, lr=0.0, eps=1e-15)  # learning rates 0: the scene stays the one the raster step is timed on
    stats = D.DensificationStats(c.P, dev)
    g = torch.Generator(device="cuda").manual_seed(1)
    gt = torch.rand((3, c.h, c.w), device=dev, generator=g)
    geo = D.DepthNormalLoss(scale_factor=0.5) if c.w_geo > 0 else None
    bg = torch.zeros(3, device=dev)  # on the device: TriangleRenderer's bg_color.to(device) is then no copy (a host-to-device copy cannot sit in a capture)
    out = {}

    def fwd_loss_bwd():
        vertex.grad = shs.grad = raw_opacity.grad = None
        pkg = D.render_view(cam, vertex, None, None, raw_opacity, shs=shs, bg_color=bg, gamma=c.gamma, active_sh_degree=c.D, max_sh_degree=c.D,
                            is_training=True, render_up_scale=c.up, rasterizer_type=c.rast)
        loss = D.photometric_loss(pkg["render"], gt, 0.8, 0.2)
        if geo is not None:
            loss = loss + c.w_geo * geo(pkg["depth"], pkg["normal"], cam.tan_fovx, cam.tan_fovy)
        if a.mimic_dense_alloc:
            dummy = torch.empty((c.P, M, 3), device=dev)  # noqa: F841  (freed at the end of this function: back in the allocator's pool like dL_dshs after the step)
        with factored_sh_grads(enabled=not a.dense_adam) as sink:
            loss.backward()
        out["factors"] = (list(sink.colors), list(sink.campos))  # tensors of this iteration (of the capture, under graph replay)
        sink.clear()
        stats.update(pkg)
        out["loss"], out["pkg"] = loss.detach(), pkg

    def step():
        if a.dense_adam:
            return opt.step()
        sk = ShGradSink()
        sk.colors, sk.campos = list(out["factors"][0]), list(out["factors"][1])
        opt.step(sh_factors=D.ShFactors(sk, vertex, c.D, shs=shs))

    def iteration():
        fwd_loss_bwd()
        step()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(5):
        iteration()
    gc.collect(); gc.disable()
    timed(iteration, 30)  # settle (clock ramp, bench.py)
    eager_ms = timed(iteration, a.iters)
    # per-kernel table of this library's kernels (HIP events on the stream), device warm
    _C.profile_reset(); _C.profile_only(""); _C.profile_enable(True)
    for _ in range(10):
        iteration()
    torch.cuda.synchronize()
    rows = {n: ms / max(k, 1) for n, ms, k in _C.profile_read()}
    _C.profile_enable(False)
    # the raster step alone: forward + backward with fixed upstream gradients at the rasterizer's resolution
    node_n = _last_num_rendered(out["pkg"])
    raster_ms = raster_step_ms(s, c, W, H, dev, a.iters)
    n_img = 3 * c.h * c.w
    nparam = c.P * (9 + 1 + 3 * M)
    V = 1
    # one launch: the SH coefficients from the factors + the vertices and opacities from their dense gradients (the vertices are read once)
    alg_sh = {"adam_step_sh_factored": c.P * (12 * V + 36 * M + 36 * M) + 28 * c.P * 10} if not a.dense_adam else {}
    alg = {"photometric_fwd": 5 * 4 * n_img, "photometric_bwd": 6 * 4 * n_img, "adam_step": 28 * nparam, "training_statistic": 68 * c.P,
           "depth_normal_fwd": 16 * c.h * c.w, "depth_normal_bwd": 32 * c.h * c.w}
    alg.update(alg_sh)
    if c.up > 1:
        per = (c.up * c.up + 1) * 4 * c.h * c.w  # per channel plane; render 3 + depth 1 + normal 3 planes in ONE launch each way (downsample_bilinear_many)
        alg["downsample_fwd"] = per * 7
        alg["downsample_bwd"] = per * 7
    raster_names = ("preprocess_fwd", "depth_census", "depth_sort", "scan", "emit_keys", "tile_sort", "tile_ranges", "render_fwd", "zero_grad_records",
                    "render_bwd", "preprocess_bwd")
    table = {}
    for name, ms in rows.items():
        e = {"avg_ms": round(ms, 4)}
        if name in alg:
            e["algorithmic_bytes_per_launch"] = int(alg[name])
            e["achieved_GBps"] = round(alg[name] / (ms * 1e-3) / 1e9, 1)
            e["hbm_frac"] = round(alg[name] / (ms * 1e-3) / 1e9 / HBM, 4)
        table[name] = e
    launches = {}
    lib_ms = sum(ms * launches.get(n, 1) for n, ms in rows.items())
    raster_lib_ms = sum(ms for n, ms in rows.items() if n in raster_names)

    # the eager loop WITHOUT the host's wait for the instance count (diff_triangle_rasterization_2D.set_instance_capacity: the sync-free forward, the
    # binning state sized from what this scene rendered x 1.3; overflow is reported through forward_overflowed()): the host runs ahead of the GPU
    pkg2d.set_instance_capacity(int(1.3 * node_n) + 4096)
    timed(iteration, 30)
    eager_sync_free_ms = timed(iteration, a.iters)
    sync_free_overflowed = bool(pkg2d.forward_overflowed()[0])
    pkg2d.set_instance_capacity(None)

    def emit(graph_ms, graph_err):
        it = graph_ms if graph_ms else eager_ms
        line = {"config": a.config, "workload": f"P={c.P}, camera {c.w}x{c.h}, render_up_scale {c.up} (raster {W}x{H}), {c.rast}, SH degree {c.D}, gamma {c.gamma:g}, "
                                                 f"L1 + SSIM{' + %.2f x depth/normal' % c.w_geo if c.w_geo else ''}, FusedAdam{'' if a.dense_adam else ' (colours from the factored gradient)'}, statistics",
                "iteration_ms_eager": round(eager_ms, 4), "iteration_ms_eager_sync_free": round(eager_sync_free_ms, 4), "sync_free_overflowed": sync_free_overflowed,
                "iteration_ms_graph": None if graph_ms is None else round(graph_ms, 4), "graph_note": graph_err,
                "raster_step_ms": round(raster_ms, 4), "iteration_minus_raster_ms": round(it - raster_ms, 4),
                "library_kernels_ms_per_iteration": round(lib_ms, 4), "of_which_rasterizer": round(raster_lib_ms, 4),
                "torch_glue_ms_per_iteration": round(max(it - lib_ms, 0.0), 4), "kernels": table}
        print(json.dumps(line), flush=True)

    emit(None, "eager only (the graph leg follows as a second line)")
    # the same iteration with render + loss + backward + statistics replayed from ONE HIP graph; the autograd graph of the last eager iteration holds
    # AccumulateGrad nodes created on the default stream -- a node that outlives its step breaks the capture: drop every reference first (bench.py)
    out.clear()
    vertex.grad = shs.grad = raw_opacity.grad = None
    gc.collect()
    try:
        gs = D.GraphedStep(fwd_loss_bwd, instance_capacity=int(1.3 * node_n) + 4096)

        factors = out["factors"]  # the capture's tensors: every replay refills them

        def graphed():
            gs.replay()
            out["factors"] = factors
            step()
        timed(graphed, 30)
        graph_ms = timed(graphed, a.iters)
        emit(graph_ms, "overflowed" if gs.overflowed()[0] else None)
    except Exception as e:
        emit(None, repr(e)[:200])
    gc.enable()


def _last_num_rendered(pkg):
    fn = pkg["render"].grad_fn
    seen = set()
    stack = [fn]
    while stack:
        f = stack.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        if hasattr(f, "num_rendered"):
            return int(f.num_rendered)
        stack.extend(n for n, _ in f.next_functions)
    raise RuntimeError("no rasterizer node in the graph")


def raster_step_ms(s, c, W, H, dev, iters):
    from diff_triangle_rasterization_2D import TriangleRasterizationSettings, center2D_sink
    if c.rast == "3D":
        from diff_triangle_rasterization_3D import TriangleRasterizer
    else:
        from diff_triangle_rasterization_2D import TriangleRasterizer
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rs = TriangleRasterizationSettings(image_width=W, image_height=H, tanfovx=s["tanfovx"], tanfovy=s["tanfovy"], viewmatrix=t(s["viewmatrix"]),
                                       projmatrix=t(s["projmatrix"]), campos=t(s["campos"]), sh_degree=c.D, gamma=c.gamma, scale_modifier=1.0,
                                       background_depth=5000.0, background=t(s["background"]), back_culling=False, rich_info=True, debug=False)
    raster = TriangleRasterizer(rs)
    vertex, shs, opacity = t(s["vertex"]).requires_grad_(True), t(s["shs"]).requires_grad_(True), t(s["opacity"]).requires_grad_(True)
    g = [t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])]

    def step():
        vertex.grad = shs.grad = opacity.grad = None
        o = raster(vertex, center2D_sink(c.P, dev), opacity, shs=shs)
        torch.autograd.backward([o[0], o[2], o[3]], g)

    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


if __name__ == "__main__":
    main()
