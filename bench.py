#!/usr/bin/env python
"""bench.py -- fwd+bwd throughput of the MI355X triangle rasterizer on BASELINE.json's headline workload.

    python bench.py [--gpus N --steps K --warmup W]            # N > 1: spawns the N ranks itself (ensure_world below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                 # the driver's form; WORLD_SIZE != --gpus is refused (exit code 2)

One "step" = one forward + one backward of the rasterizer over one view of the synthetic scene
S(P=1M, 1920x1080, SH degree 3, rich_info=True) (SURVEY.md 8d / BASELINE.md 4), through the drop-in Python
package, i.e. through the C ABI of libts2d.so.  Inputs are resident in HBM before the timed region.  With N > 1
every rank renders its own view of the same triangles (image-parallel, weak scaling) and the per-triangle
gradients are exchanged over RCCL (reduce-scatter + all-gather of one bucket on a side stream, waited for INSIDE the step that produced
them -- north_star's all-reduce semantics; --delayed-exchange collects them one step later instead).  Rank 0 prints
ONE JSON line.

Sequence: one cold step, two steps that find the dominant kernel, garbage collection (then off), --settle-steps + W untimed steps identical
to the timed ones with NOTHING between the last of them and the barrier that opens the timed region (the device needs ~25 ms of load to
bring its clocks back after an idle gap; config.settle_steps_untimed), exactly K steps between barriers + synchronize, then ten steps with
every kernel bracketed by HIP events (kernels_avg_ms).  Every timed step leaves ONE event on the launch stream: config.device_step_ms is the
spread of the time between consecutive events (what the GPU saw), config.host_step_ms the spread of the host time per queued step,
config.gpu_idle_ms_per_step the mean device step minus the sum of the kernels' own durations -- a stall is attributable to the host or to
the device from these.

Extra objects in the JSON line:
  roofline     -- the dominant kernel's algorithmic bytes / its average duration (HIP events on the launch stream,
                  recorded inside the timed region by libts2d's profile hook) against the 8 TB/s HBM peak.
  cpu_baseline -- the CPU oracle (oracle/, OpenMP, all host cores) timed once on the same scene (N=1, rank 0 only).
The oracle is used here only as the timed CPU baseline and to cross-check the image; it is never on the product path.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def msb_bits(ntiles: int) -> int:
    n, b = ntiles, 0
    while n:
        n >>= 1
        b += 1
    return b


def algorithmic_bytes(P, N, W, H, D, ntiles):
    """per-kernel algorithmic HBM bytes of one fwd+bwd step (rich_info, SH colour)."""
    sh = 12 * (D + 1) ** 2
    passes = -(-(32 + msb_bits(ntiles)) // 8)
    k = {
        "preprocess_fwd": P * (36 + sh + 99),
        "scan": P * 8,
        "emit_keys": P * 28 + N * 12,
        "sort_pairs": N * 24 * passes,  # the reference's single 64-bit sort; counted in `total`
        "tile_ranges": N * 8,
        "render_fwd": N * 72 + W * H * 36,
        "render_bwd": N * 72 + W * H * 36 + P * 2 * 64,
        "preprocess_bwd": P * (36 + sh + 3 + 4 + 48 + 12) + P * (36 + 8 + sh),
    }
    k["total"] = sum(k.values())
    # our two-stage ordering (binning.hip) moves fewer bytes than the SURVEY row; listed for the per-kernel table only
    k["depth_sort"] = P * 16 * 4
    k["depth_census"] = P * 8          # first histogram: keys + tile counts (small scenes: the whole depth order in this one launch)
    k["zero_grad_records"] = P * 64
    k["preprocess_colour"] = P * (4 + 36 + sh + 13)   # the SH colours on the side stream: tile count + vertex + SH row in, r g b + clamp flags out
    k["tile_sort"] = N * 16 * -(-msb_bits(ntiles) // 8)
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--triangles", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--edge-px", type=float, default=6.0, help="mean projected triangle edge in pixels (SURVEY 8d: 6 = the headline; anything else is NOT the headline)")
    ap.add_argument("--gamma", type=float, default=1.0,
                    help="the reference's compactness exponent (scheduled 1 -> 50 during training, VanillaTS_model.py:549-554); 1 = the headline "
                         "(SURVEY 8d), anything else is NOT the headline: the blend kernels then evaluate ecc^(2 gamma) with a log / exp pair")
    ap.add_argument("--scene-mode", default="frustum", choices=["frustum", "centered", "maincu"],
                    help="frustum = SURVEY 8d's uniform scene (the headline); centered = the same triangles concentrated about the optical axis "
                         "(object-centric view, non-uniform load over the tiles); NOT the headline when changed")
    ap.add_argument("--settle-steps", type=int, default=40,
                    help="untimed steps in front of the W warm-up steps (0 = none): ~70 ms of load bring the device's clocks up; what matters is that "
                         "NOTHING sits between the last warm-up step and the timed region (see the sequence comment in main)")
    ap.add_argument("--rasterizer", default="2D", choices=["2D", "3D"],
                    help="2D = the headline path (BASELINE.json); 3D = the ray/plane variant (SURVEY.md 8f rank 1), not the headline")
    ap.add_argument("--dense-exchange", action="store_true",
                    help="N > 1: all-reduce the dense dL_dshs (60 floats/triangle) instead of the factored exchange (15 + 3 per view)")
    ap.add_argument("--delayed-exchange", action="store_true",
                    help="N > 1: double-buffered buckets, the exchange of step i is waited for after step i+1's kernels are queued (one-step-delayed "
                         "application of the gradients -- NOT north_star's semantics); default: synchronous, step i's reduced gradients are waited "
                         "for before step i+1's forward is queued")
    ap.add_argument("--sync-exchange", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--exchange-compare", action="store_true", help="(default for N > 1 since round 6; kept for old command lines)")
    ap.add_argument("--no-exchange-compare", action="store_true",
                    help="N > 1: by default K more steps are timed in the OTHER exchange mode behind the timed region and reported beside the headline "
                         "mode's figures (config.exchange.other_mode: ms_per_step, exposed_ms_per_step) -- one multi-GPU lease yields both exposure "
                         "figures; this switches that leg off.  The headline `value` is never affected: the leg runs after the timed region")
    ap.add_argument("--with-optimizer", action="store_true",
                    help="NOT the headline metric: every step also runs the fused Adam step (diff_recon_hip.FusedAdam: vertex, opacity and the SH tensor "
                         "with the reference's f_dc / f_rest learning-rate split, one launch) on the step's gradients, with all learning rates 0 so that "
                         "the scene stays the one BASELINE.json names; reported as config.optimizer")
    ap.add_argument("--depth-lsd", action="store_true", help="lab library only: the LSD depth sort of rounds 2-5 at every size instead of the sampled-splitter "
                    "form (binning.hip: depth_split_*), for A/B runs")
    ap.add_argument("--depth-split-all", action="store_true", help="lab library only: the sampled-splitter depth order up to the 1.6 M triangles it supports "
                    "(the product switches to the LSD sort above 500 000), for A/B runs")
    ap.add_argument("--force-depth-pass4", action="store_true",
                    help="NOT the headline: the depth sort runs its fourth pass although every depth of the synthetic scene shares the top key byte "
                         "(what a scene spanning more than a factor of four in depth costs).  Needs the lab library: "
                         "TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so (the product library has no switch)")
    ap.add_argument("--side-stream", action="store_true",
                    help="NOT the headline, lab library only (TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so): the SH colours evaluated by a throttled kernel on a "
                         "library-owned stream beside the ordering chain (csrc/api.hip: SideLane) -- round 6's measured negative result, kept reproducible "
                         "(profiles/r06_side_stream.txt)")
    ap.add_argument("--colour-blocks", type=int, default=0, help="with --side-stream: the colour kernel's grid = its throttle (default 512)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--timed-kernel-events", default="dominant", choices=["dominant", "all"],
                    help="which kernels are bracketed by HIP events INSIDE the timed region: the dominant one (2 events per step; default) or "
                         "every kernel (measurement of the kernels under sustained load; costs a few per cent of `value`)")
    ap.add_argument("--dump-steps", action="store_true", help="config.device_step_ms_all / host_step_ms_all: the per-step sequences, in order")
    ap.add_argument("--sync-free", action="store_true",
                    help="use the sync-free forward (ts2d_forward, capacity = 1.25 x the instance count of the cold step) instead of the "
                         "reference's sequence with its blocking read of num_rendered; overflow is checked after the timed region")
    ap.add_argument("--sparse-exchange", action="store_true",
                    help="N > 1: only the gradient rows of triangles SOME rank saw this step travel (parallel.VisibleRows: a MAX all-reduce of P bytes "
                         "behind the forward, compact bucket, compact colour factors).  Pays where a view sees a fraction of the scene (BASELINE "
                         "configs[4]); on the synthetic headline scene every triangle is in the frustum (config.exchange.visible_fraction ~ 1) and "
                         "the dense exchange -- the default -- moves the same bytes without the two gathers")
    ap.add_argument("--range-exchange", type=int, default=0, metavar="K",
                    help="N > 1: the backward's per-triangle kernel runs as K launches over consecutive triangle ranges and the bucket is all-reduced "
                         "range by range as they finish (GradBucket.prepare_ranges / reduce_ranges_async, ts2d_backward_ranged): what it can hide is "
                         "bounded by that kernel (0.107 ms of the 1.59 ms step); compare config.exchange.exposed_ms_per_step with the default's")
    ap.add_argument("--forward-only", action="store_true",
                    help="NOT the headline: the evaluation / viewer path -- forward with rich_info=False under no_grad, no backward "
                         "(VanillaTS_trainer.py:167, viser_viewer.py:199-229); `metric` names itself, `value` = forward Mpix/s")
    ap.add_argument("--hip-graph", action="store_true",
                    help="NOT the driver's command: the step (sync-free forward, loss gradients, backward) is captured ONCE into a HIP graph "
                         "(torch.cuda.CUDAGraph on the rasterizer's launches; the sync-free forward has no host read to break the capture) and the "
                         "timed steps are graph replays -- what a training loop with a fixed triangle count between densifications can do.  Host "
                         "cost per step = one graph launch; implies --sync-free, N = 1")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="NOT a measurement: after the ranks have been created (see ensure_world) every rank joins the process group, the ranks "
                         "all-gather their (rank, pid) and rank 0 prints them as one JSON line -- what tests/test_bench_launch_cpu.py runs over gloo "
                         "to pin that `--gpus N` alone produces N ranks; needs no HIP device")
    args = ap.parse_args()
    if args.hip_graph:
        args.sync_free = True
    args.exchange_compare = args.gpus > 1 and not args.no_exchange_compare

    rank, local_rank, world = ensure_world(args)
    if args.rendezvous_check:
        return rendezvous_check(rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the rasterizer has no CPU fallback)")
    # TS2D_BENCH_BACKEND=gloo is a functional check of the N > 1 code path on a box with fewer GPUs than ranks (ranks
    # then share devices and tensors travel through the host); measurements use the default, RCCL.
    backend = os.environ.get("TS2D_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev if backend == "nccl" else None)

    import synthetic
    from diff_triangle_rasterization_2D import TriangleRasterizationSettings, TriangleRasterizer, _C, center2D_sink
    if args.rasterizer == "3D":
        from diff_triangle_rasterization_3D import TriangleRasterizer
    from diff_triangle_rasterization_2D import parallel
    from diff_triangle_rasterization_2D.parallel import GradBucket

    if args.force_depth_pass4:
        if not hasattr(_C._lib, "ts2d_lab_force_depth_pass4"):
            raise SystemExit("--force-depth-pass4 needs TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so")
        _C._lib.ts2d_lab_force_depth_pass4(1)
    if args.depth_lsd:
        if not hasattr(_C._lib, "ts2d_lab_depth_split"):
            raise SystemExit("--depth-lsd needs TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so")
        import ctypes
        _C._lib.ts2d_lab_depth_split.argtypes = [ctypes.c_int, ctypes.c_int]
        _C._lib.ts2d_lab_depth_split(1, 0)
    if args.depth_split_all:
        if not hasattr(_C._lib, "ts2d_lab_depth_split"):
            raise SystemExit("--depth-split-all needs TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so")
        import ctypes
        _C._lib.ts2d_lab_depth_split.argtypes = [ctypes.c_int, ctypes.c_int]
        _C._lib.ts2d_lab_depth_split(2, 0)
    if args.side_stream:
        if not hasattr(_C._lib, "ts2d_lab_side_stream"):
            raise SystemExit("--side-stream needs TS2D_LIBRARY_PATH=tools/bin/libts2d_lab.so")
        _C._lib.ts2d_lab_side_stream(1)
        if args.colour_blocks:
            _C._lib.ts2d_lab_colour_blocks(args.colour_blocks)
    P, W, H, D = args.triangles, args.width, args.height, args.sh_degree
    s = synthetic.scene(P, W, H, D, seed=42, mode=args.scene_mode, edge_px=args.edge_px)
    # one view per rank: same triangles, camera shifted sideways by a few world units per rank
    cam = synthetic.camera(W, H)
    if rank > 0:
        view = cam["viewmatrix"].copy()
        shift = np.array([7.0 * rank, -3.0 * rank, 0.0], np.float32)
        view[3, :3] -= shift * np.array([-1, 1, -1], np.float32)  # translate the camera centre by `shift`
        cam["viewmatrix"] = view
        cam["projmatrix"] = (view @ synthetic.projection_matrix(cam["tanfovx"], cam["tanfovy"]).T).astype(np.float32)
        cam["campos"] = np.array([0, 0, synthetic.CAM_DIST], np.float32) + shift
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rs = TriangleRasterizationSettings(
        image_width=W, image_height=H, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], viewmatrix=t(cam["viewmatrix"]),
        projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"]), sh_degree=D, gamma=args.gamma, scale_modifier=1.0,
        background_depth=5000.0, background=t(s["background"]), back_culling=False, rich_info=not args.forward_only, debug=False)
    raster = TriangleRasterizer(rs)
    vertex = t(s["vertex"]).requires_grad_(True)
    shs = t(s["shs"]).requires_grad_(True)
    opacity = t(s["opacity"]).requires_grad_(True)
    g_feat, g_depth, g_norm = t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])
    bucket = None
    factored = world > 1 and not args.dense_exchange
    M = shs.shape[1]
    buckets, shx, wait_events = [], [], []
    overlap = world > 1 and args.delayed_exchange
    two_buckets = world > 1 and (overlap or args.exchange_compare)
    if world > 1:
        # (no slot for dL_dcenter2D: it is a per-view statistic, not a parameter gradient -- each view keeps its own, nothing to sum)
        shapes = [vertex.shape, opacity.shape] + ([] if factored else [shs.shape])
        # two process groups = two RCCL communicators / streams: the bucket's reduce-scatter + all-gather and the SH-gradient all-gather
        # are in flight together; two buckets: the exchange of step i has the whole of step i + 1 to finish
        bucket_group, sh_group = parallel.exchange_groups()
        for _ in range(2 if two_buckets else 1):
            buckets.append(GradBucket(shapes, dev, group=bucket_group, names=["vertex", "opacity"] + ([] if factored else ["color"])))
            if args.range_exchange > 1:
                if args.sparse_exchange:
                    raise SystemExit("--range-exchange and --sparse-exchange are alternatives")
                buckets[-1].prepare_ranges(args.range_exchange)
            shx.append(parallel.FactoredShExchange(sh_group, dev))
        bucket = buckets[0]
    sink = parallel.ShGradSink()
    vis_rows = parallel.VisibleRows(bucket_group if world > 1 else None, dev) if (world > 1 and args.sparse_exchange) else None

    state = {"step": 0, "overlap": overlap, "pending": None}
    optimizer = None
    if args.with_optimizer:
        if world > 1:
            raise SystemExit("--with-optimizer is a single-GPU measurement (the sharded form is diff_recon_hip.ShardedAdam)")
        from diff_recon_hip import FusedAdam
        optimizer = FusedAdam([{"params": [vertex], "lr": 0.0, "name": "vertex"}, {"params": [opacity], "lr": 0.0, "name": "opacity"},
                               {"params": [shs], "lr": 0.0, "lr_tail": 0.0, "tail_period": 3 * M, "tail_split": 3, "name": "shs"}], lr=0.0, eps=1e-15)

    def collect(i):
        """Waits (on the compute stream) for the exchange that step i started; the wait is bracketed by events = the EXPOSED exchange time."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        state["grads"] = buckets[i % len(buckets)].wait()
        if factored:
            state["shs_grad"] = shx[i % len(shx)].wait()  # dense dL_dshs summed over all ranks' views (one each)
        e1.record()
        wait_events.append((e0, e1))

    def step():
        center2D = center2D_sink(P, dev)  # the caller's gradient sink (triangle_renderer.py:67), as diff_recon_hip.TriangleRenderer makes it: a fresh leaf per step, no fill kernel
        if bucket is not None:
            # N > 1: the backward kernels write dL_dvertex / dL_dopacity (and, with --dense-exchange, dL_dshs)
            # straight into the exchange bucket; its reduce-scatter + all-gather starts on a side stream as soon as the backward
            # is queued, the factored SH-gradient all-gather + expansion on another (parallel.py)
            i = state["step"]
            b, x = buckets[i % len(buckets)], shx[i % len(shx)]
            with b.capture(), parallel.factored_sh_grads(sink, enabled=factored):
                out = raster(vertex, center2D, opacity, shs=shs)
                if vis_rows is not None:
                    vis_rows.begin([out[1]])  # the union of radii > 0 over the ranks: queued behind the forward, read after the backward is queued
                torch.autograd.backward([out[0], out[2], out[3]], [g_feat, g_depth, g_norm])
            if args.range_exchange > 1:
                b.reduce_ranges_async()
            else:
                b.reduce_async(rows=vis_rows)
            if factored:
                x.start(sink, vertex, D, M, uniform=True, rows=vis_rows)
            if vis_rows is not None:
                state["visible_rows"] = int(vis_rows.index().numel())
            prev, state["pending"] = state["pending"], i
            if not state["overlap"]:
                collect(i)       # synchronous (default): this step's reduced gradients before anything of the next step is queued
                state["pending"] = None
            elif prev is not None:
                collect(prev)    # one-step-delayed application: the previous step's gradients arrive while this step's exchange is in flight
            state["step"] = i + 1
        elif args.forward_only:
            with torch.no_grad():
                out = raster(vertex, center2D, opacity, shs=shs)
            state["image"] = out[0]
            return
        else:
            out = raster(vertex, center2D, opacity, shs=shs)
            torch.autograd.backward([out[0], out[2], out[3]], [g_feat, g_depth, g_norm])
        if optimizer is not None:
            optimizer.step()
        state["num_rendered"] = out[0].grad_fn.num_rendered
        state["image"] = out[0]
        vertex.grad = None
        shs.grad = None
        opacity.grad = None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Sequence (round 4; profiles/r04_notes.md has the per-step device times behind it):
    #   1. one cold step (library initialisation, allocator growth), then two steps with EVERY kernel bracketed by HIP events -- only to
    #      find the dominant kernel, the one the timed region brackets (2 events per step; all-kernel events cost a few per cent);
    #   2. garbage collection now, the collector off until the timed region is over (like a training loop would);
    #   3. --settle-steps untimed steps + the W warm-up steps, identical to the timed ones, and NOTHING between the last of them and the
    #      barrier that opens the timed region.  The MI355X takes ~25 ms of load (~15 steps) to bring its clocks back after an idle gap of
    #      some tens of milliseconds (1.62 ms/step steady, 2.2 ms for the first step after such a gap): rounds 2 and 3 ran their warm-up
    #      steps, then read event tables and collected garbage, and started the clock on a device that had just dropped its clocks;
    #   4. the K timed steps between barriers; 5. ten more steps with every kernel bracketed = the per-kernel table, device still warm.
    # A fixed COUNT of settle steps, not a duration: with N > 1 every step contains collectives, so all ranks run the same number.
    events = not args.no_kernel_events
    step()  # cold step -- never timed
    barrier()
    if args.sync_free:
        import diff_triangle_rasterization_2D as _pkg
        _pkg.set_instance_capacity(int(1.25 * int(state["num_rendered"])) + 1024)
        step()
        barrier()
    dominant = ""
    if events:
        _C.profile_reset()
        _C.profile_only("")
        _C.profile_enable(True)
        for _ in range(2):
            step()
        barrier()
        rows0 = _C.profile_read()
        dominant = max(rows0, key=lambda r: r[1] / max(r[2], 1))[0] if rows0 else ""
        _C.profile_reset()
        _C.profile_only(dominant if args.timed_kernel_events == "dominant" else "")
    eager_step = step
    if args.hip_graph:
        if world > 1:
            raise SystemExit("--hip-graph is a single-GPU measurement")
        if events:
            _C.profile_enable(False)  # the profile hook records HIP events: not inside a capture
        # the autograd graph of the last eager step (kept alive by state["image"]) holds AccumulateGrad nodes created on the DEFAULT stream; a node
        # that outlives its step is reused by the next one, and a default-stream node inside a capture is an illegal dependency: drop it first
        state["image"] = None
        vertex.grad = shs.grad = opacity.grad = None
        gc.collect()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # torch's capture recipe: a few eager iterations on the side stream first
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        torch.cuda.synchronize()
        step = graph.replay
        if events:
            _C.profile_enable(True)
    gc.collect()
    gc.disable()
    settle_steps = max(args.settle_steps, 0)
    for _ in range(settle_steps + max(args.warmup, 0)):
        step()
    if events:
        _C.profile_reset()  # drains the pending events: the device idles for microseconds, not for a table read

    def timed_region(delayed):
        """Exactly K steps between barriers; returns (elapsed s, host stamps, per-step device events, exposed exchange ms per step)."""
        if state["pending"] is not None:  # delayed mode: the exchange of the last step before this region is still out
            collect(state["pending"])
            state["pending"] = None
        state["overlap"] = delayed
        wait_events.clear()
        stamps, marks = [], [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        barrier()
        t_start = time.perf_counter()
        marks[0].record()
        for k in range(args.steps):
            step()
            marks[k + 1].record()               # ONE event per step on the launch stream: see `device_step_ms`
            stamps.append(time.perf_counter())  # host time at which the step was queued (no synchronisation added): see `host_step_ms`
        if state["pending"] is not None:
            collect(state["pending"])  # delayed mode: the last step's exchange belongs to the timed region
            state["pending"] = None
        barrier()
        took = time.perf_counter() - t_start
        exposed = sum(a.elapsed_time(b) for a, b in wait_events) / max(args.steps, 1) if wait_events else None
        return took, [t_start] + stamps, marks, exposed

    elapsed, stamps, marks, exposed_ms = timed_region(overlap)
    timed_rows = _C.profile_read() if events else []
    other = None
    if world > 1 and args.exchange_compare:
        if events:
            _C.profile_enable(False)
        o_elapsed, _, _, o_exposed = timed_region(not overlap)
        other = {"mode": "delayed" if not overlap else "synchronous", "ms_per_step": o_elapsed, "exposed_ms_per_step": o_exposed}
    warm_rows = []
    if events and world == 1:
        # the per-kernel table: ten more steps with every kernel bracketed, right behind the timed region (device warm)
        _C.profile_reset()
        _C.profile_only("")
        _C.profile_enable(True)
        for _ in range(10):
            eager_step()  # (--hip-graph: a replay carries no per-kernel events; the table is the eager step's)
        barrier()
        warm_rows = _C.profile_read()
    gc.enable()
    t0 = stamps[0]
    stamps = stamps[1:]
    device_seq = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    device_steps = sorted(device_seq)
    if events:
        _C.profile_enable(False)
        _C.profile_only("")
    if world > 1:
        tmax = torch.tensor([elapsed, other["ms_per_step"] if other else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        if other:
            other["ms_per_step"] = round(1e3 * float(tmax[1].item()) / args.steps, 4)
            other["exposed_ms_per_step"] = None if other["exposed_ms_per_step"] is None else round(other["exposed_ms_per_step"], 4)

    if args.forward_only:  # no autograd node: count the instances with one more forward that keeps its graph
        out = raster(vertex, center2D_sink(P, dev), opacity, shs=shs)
        state["num_rendered"] = out[0].grad_fn.num_rendered
    true_n = int(state["num_rendered"])
    if args.sync_free:
        over, true_n = _pkg.forward_overflowed(state["image"])
        if over:
            raise SystemExit("sync-free forward overflowed its instance capacity: the timed steps rendered nothing")
    ms_per_step = 1e3 * elapsed / args.steps
    host_gaps = sorted(1e3 * (b - a) for a, b in zip([t0] + stamps[:-1], stamps))
    mpix_s = world * W * H / (elapsed / args.steps) / 1e6
    N = true_n
    ntiles = ((W + 15) // 16) * ((H + 15) // 16)
    alg = algorithmic_bytes(P, N, W, H, D, ntiles)
    if args.forward_only:  # 48 B per instance and 20 B per pixel without rich_info; no backward rows
        alg["render_fwd"] = N * 48 + W * H * 20
        alg["total"] = alg["preprocess_fwd"] + alg["scan"] + alg["emit_keys"] + alg["sort_pairs"] + alg["tile_ranges"] + alg["render_fwd"]

    result = {
        # BASELINE.json's metric is quoted on the headline configuration; any other --triangles / --width / --height names itself
        "metric": (f"forward-only (rich_info=False, evaluation path) Mpixels/sec @ {P} triangles, {W}x{H}" if args.forward_only else
                   "fwd+bwd Mpixels/sec @ 1M triangles, 1920x1080; achieved HBM GB/s" if (P, W, H) == (1_000_000, 1920, 1080)
                   else f"fwd+bwd Mpixels/sec @ {P} triangles, {W}x{H}; achieved HBM GB/s"),
        "value": round(mpix_s, 3), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"rasterizer": args.rasterizer, "workload": (f"S(P={P}, {W}x{H}, SH degree {D}, rich_info=False, gamma={args.gamma:g}): forward of one view (evaluation path)" if args.forward_only else
                                                               f"S(P={P}, {W}x{H}, SH degree {D}, rich_info, gamma={args.gamma:g}): fwd+bwd of one view per GPU"),
                   "triangles": P, "width": W, "height": H, "sh_degree": D, "num_rendered": N, "scene_mode": args.scene_mode, "edge_px": args.edge_px,
                   "forward": ("sync-free, the whole step replayed from ONE captured HIP graph (torch.cuda.CUDAGraph): not the driver's command" if args.hip_graph else
                               "sync-free (ts2d_forward, device-side instance count)" if args.sync_free else
                               "speculative (ts2d_forward_speculative: queued for 1.25 x the recent instance count, exact num_rendered read back "
                               "behind the queue; the package default)"),
                   "parallelism": f"image-parallel x{world}" + ((", RCCL reduce-scatter + all-gather of one 10-float/triangle bucket the backward writes into (GradBucket.capture) + all-gather of factored SH grads (3 floats/triangle/view)" if factored
                                       else ", RCCL all-reduce of dense per-triangle grads (60 floats/triangle)") if world > 1 else ""),
                   "exchange": ({"mode": "delayed: double-buffered buckets, step i's exchange is waited for behind step i+1's kernels (one-step-delayed application)"
                                         if overlap else "synchronous: step i's reduced gradients are waited for inside step i (north_star's all-reduce semantics)",
                                 "exposed_ms_per_step": None if exposed_ms is None else round(exposed_ms, 4),
                                 "rows": ("visible rows only (parallel.VisibleRows)" if vis_rows is not None else
                                          f"all rows, in {args.range_exchange} triangle ranges behind the ranged backward" if args.range_exchange > 1 else "all rows (dense)"),
                                 "visible_fraction": (round(state.get("visible_rows", 0) / max(P, 1), 4) if vis_rows is not None else None),
                                 "bucket_bytes_per_rank_and_step": (buckets[0].last_exchanged_bytes if buckets and hasattr(buckets[0], "last_exchanged_bytes") else None),
                                 "wire_bytes_per_rank_and_step": wire_bytes(world, buckets[0].last_exchanged_bytes if buckets and hasattr(buckets[0], "last_exchanged_bytes") else 0,
                                                                            P * 12 * (state.get("visible_rows", P) / max(P, 1) if vis_rows is not None else 1.0) if factored else 0,
                                                                            exposed_ms if not overlap else None),
                                 "other_mode": other, "process_groups": 2} if world > 1 else None),
                   "distributed": backend_info(backend, world),
                   # spread of the host-side time per queued step: a stalled host (allocator growth, garbage collection, a descheduled thread)
                   # shows up here as a maximum far above the median, and in `value` (the contract times all K steps, stalls included)
                   "settle_steps_untimed": settle_steps,
                   "depth_sort_passes": "4 (forced, lab library)" if args.force_depth_pass4 else "3 or 4, decided on the device by the key-bit census (3 on this scene)",
                   "optimizer": ("fused Adam step inside every step (vertex, opacity, SH with two learning rates; learning rates 0): NOT the headline metric"
                                 if optimizer is not None else None),
                   "host_step_ms": {"min": round(host_gaps[0], 3), "median": round(host_gaps[len(host_gaps) // 2], 3), "max": round(host_gaps[-1], 3)},
                   # time between the per-step events on the launch stream: what the GPU saw.  A device step far above the median with an
                   # unremarkable host step is a device-side stall; both high together = the host starved the queue
                   "device_step_ms": {"min": round(device_steps[0], 3), "median": round(device_steps[len(device_steps) // 2], 3),
                                      "max": round(device_steps[-1], 3), "mean": round(sum(device_steps) / len(device_steps), 4)},
                   "algorithmic_bytes_per_step": alg["total"],
                   "achieved_hbm_gbs_whole_step": round(alg["total"] / (elapsed / args.steps) / 1e9, 2),
                   "hbm_roofline_frac_whole_step": round(alg["total"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)},
    }

    if args.dump_steps:
        result["config"]["device_step_ms_all"] = [round(x, 3) for x in device_seq]
        result["config"]["host_step_ms_all"] = [round(1e3 * (b - a), 3) for a, b in zip([t0] + stamps[:-1], stamps)]
    if rank == 0:
        if events:
            rows = timed_rows  # the dominant kernel (or, with --timed-kernel-events all, every kernel), timed inside the timed region
            if not rows:       # --hip-graph: replays carry no events; the dominant kernel's duration is the eager table's
                rows = [r for r in warm_rows if r[0] == dominant]
            if args.timed_kernel_events == "all":
                result["kernels_avg_ms_timed_region"] = {name: round(ms / max(n, 1), 4) for name, ms, n in rows}
                result["config"]["gpu_busy_ms_per_step_timed_region"] = round(sum(ms / max(n, 1) for _, ms, n in rows), 4)
                rows = [r for r in rows if r[0] == dominant]
            kernels = {name: ms / max(n, 1) for name, ms, n in warm_rows}
            result["kernels_avg_ms"] = {k: round(v, 4) for k, v in kernels.items()}  # ten steps right behind the timed region
            if world == 1:
                busy = sum(kernels.values())
                side = kernels.get("preprocess_colour", 0.0) if args.side_stream else 0.0  # (lab experiment: a kernel BESIDE the main chain)
                result["config"]["gpu_busy_ms_per_step"] = round(busy, 4)  # sum of the kernels' own durations
                result["config"]["gpu_idle_ms_per_step"] = round(sum(device_steps) / len(device_steps) - (busy - side), 4)
            dom, dom_ms, dom_n = rows[0]
            dom_avg = dom_ms / max(dom_n, 1)
            ach = alg.get(dom, 0) / (dom_avg * 1e-3) / 1e9
            headline = (P, W, H, D, args.rasterizer) == (1_000_000, 1920, 1080, 3, "2D") and not args.forward_only
            result["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": hbm_traffic(dom) if headline else None,
                                  "algorithmic_bytes_per_launch": alg.get(dom, 0), "avg_launch_ms": round(dom_avg, 4),
                                  "launches_timed": dom_n, "compute": compute_side(dom, dom_avg, headline),
                                  "counter_source": counter_source() if headline else None,
                                  "note": "the blend kernels are bound by VALU issue, not by HBM (DESIGN.md section 4): `compute` is the "
                                          "roofline they sit on; every duration is this run's timed-region HIP events"}
        else:
            result["roofline"] = None
        if world == 1 and not args.no_cpu_baseline and not args.forward_only:
            result["cpu_baseline"] = cpu_baseline(s, state["image"].detach().cpu().numpy(), 3 if args.rasterizer == "3D" else 2)
            ref = reference_gpu_baseline(s, dev, args.rasterizer, state["image"].detach())
            if ref is not None:
                result["reference_gpu"] = ref
                result["reference_gpu"]["speedup"] = round(result["value"] / ref["value"], 2)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def ensure_world(args):
    """`--gpus N` IS the number of ranks.  Three cases:
      * WORLD_SIZE set (the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): it must equal --gpus, anything else
        is a launch error -- exit code 2, nothing is measured (a line that said "n_gpus": 1 for a command that asked for 8 would be worse than none);
      * WORLD_SIZE unset and --gpus 1: this process is the one rank;
      * WORLD_SIZE unset and --gpus N > 1 (the N = 1 command with the number changed): this process becomes the LAUNCHER -- it re-executes the
        same command line under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1, a free port), forwards the children's
        output and exit code, and never touches a device itself.
    Returns (rank, local_rank, world) in a process that is a rank."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                             f"(python bench.py --gpus N spawns them itself; under torch.distributed.run use --nproc-per-node == --gpus)\n")
            raise SystemExit(2)
        return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(env_world)
    if args.gpus <= 1:
        if args.gpus < 1:
            raise SystemExit("--gpus must be >= 1")
        return 0, 0, 1
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a free port for the rendezvous (the driver passes its own through torchrun)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    raise SystemExit(subprocess.call(cmd, env=env))


def wire_bytes(world, bucket_bytes, sh_factor_bytes, exposed_ms):
    """What one rank must SEND per step for the exchange it was handed (the figure a link-level counter would be compared with): reduce-scatter +
    all-gather of the bucket = 2 (N - 1) / N of its bytes, the all-gather of the colour factors = (N - 1) x this rank's factors; and, in the
    synchronous mode (the whole exchange is exposed), the bus rate that the exposed time implies.  xGMI: 7 links x ~153 GB/s per GPU."""
    if world <= 1:
        return None
    bucket = 2.0 * (world - 1) / world * bucket_bytes
    sh = (world - 1) * sh_factor_bytes
    out = {"bucket_rs_ag": int(bucket), "sh_factors_all_gather": int(sh), "total": int(bucket + sh),
           "expected_ms_at_7x153_GBs_all_links": round((bucket + sh) / (7 * 153e9) * 1e3, 4),
           "expected_ms_ring_one_link_153_GBs": round((bucket + sh) / 153e9 * 1e3, 4)}
    if exposed_ms:
        out["implied_bus_GBs_from_exposed_ms"] = round((bucket + sh) / (exposed_ms * 1e-3) / 1e9, 2)
    return out


def backend_info(backend, world):
    """What the exchange ran on, for the JSON line: torch.distributed's backend name and, over RCCL, the library's version."""
    info = {"world": world, "backend": (dist.get_backend() if world > 1 else None), "requested": backend if world > 1 else None}
    if world > 1 and backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            info["rccl_version"] = None
    return info


def rendezvous_check(rank, world):
    """--rendezvous-check: every rank reports (rank, pid); rank 0 prints what it gathered.  gloo unless TS2D_BENCH_BACKEND says otherwise, no device."""
    backend = os.environ.get("TS2D_BENCH_BACKEND", "gloo")
    seen = [(rank, os.getpid())]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        seen = [None] * world
        dist.all_gather_object(seen, (rank, os.getpid()))
    if rank == 0:
        print(json.dumps({"rendezvous_check": True, "n_gpus": world, "ranks": sorted(r for r, _ in seen), "distinct_processes": len({p for _, p in seen}),
                          "distributed": backend_info(backend, world)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


PMC_NAMES = {"render_fwd": "render_fwd_group_kernel<rich>", "render_bwd": "render_bwd_group_kernel<rich>"}
TRAFFIC_NAMES = {"render_fwd": "render_fwd_group", "render_bwd": "render_bwd_group", "emit_keys": "scan_emit", "scan": "gather_blocksum"}
USEFUL_FLOP_PER_PAIR = {"render_fwd": 60, "render_bwd": 175}  # fp32 operations a blended (pixel, triangle) pair needs
FP32_VECTOR_PEAK = 157.3e12                                   # MI355X_MICROARCH.md


def compute_side(kernel, avg_ms, headline):
    """The roofline the blend kernels actually sit on (they are bound by VALU issue), from the committed SQ-counter
    pass of the same workload (profiles/blend_pmc.json, tools/collect_blend_pmc.sh), the static instruction mix of the kernel's step
    loop (profiles/r06_valu_mix.json, tools/isa_mix.py) and the lane-group statistics (profiles/blend_stats.json).  Two bounds on the
    VALU time, both in the kernel's own cycles (GRBM_GUI_ACTIVE of the launch):
      valu_busy_frac_lower = SQ_INSTS_VALU x 2 cycles (every instruction at the full wave64-on-SIMD32 rate) / (SIMDs x cycles);
      valu_busy_frac_upper = SQ_INSTS_VALU x the step loop's mix priced with the issue costs measured in REAL shader cycles
                             (tools/valu_bench3.hip: 2 / 4 / 8 cycles for full-rate / half-rate / transcendental instructions; half-rate
                             instructions interleaved with FMAs issue faster than that, hence an upper estimate).
    Plus blended (pixel, triangle) pairs per second and the useful fp32 rate as a fraction of the 157 TFLOP/s vector peak.
    Durations: the timed-region HIP events of THIS run; everything counter-derived comes from the committed passes named in
    profiles/counters_manifest.json (copied into the line as roofline.counter_source)."""
    out = {}
    if not headline:  # the counter passes were collected on the 2D headline workload only: nothing to quote for another one
        return None
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "blend_pmc.json"))).get(PMC_NAMES.get(kernel, ""))
        if v:
            out.update(valu_insts_per_launch=v["SQ_INSTS_VALU"], valu_busy_frac_lower=v["valu_issue_frac_at_2cyc"],
                       transcendental_insts_per_launch=v.get("SQ_INSTS_VALU_TRANS_F32"), lds_busy_frac=v.get("lds_busy_frac"),
                       avg_waves_per_simd=v.get("avg_waves_per_simd"), wave_cycle_split=v.get("wave_cycle_split"),
                       kernel_cycles=v.get("kernel_cycles"))
            mix = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_mix.json"))).get(kernel)
            if mix and v.get("kernel_cycles"):
                out.update(step_loop_mix={k: mix[k] for k in ("full", "half", "trans")},
                           priced_cycles_per_valu_instruction=mix["priced_cycles_per_valu_instruction"],
                           valu_busy_frac_upper=round(min(1.0, v["SQ_INSTS_VALU"] * mix["priced_cycles_per_valu_instruction"] / (1024.0 * v["kernel_cycles"])), 4))
    except Exception:
        pass
    try:
        st = json.load(open(os.path.join(ROOT, "profiles", "blend_stats.json")))
        if headline and kernel in USEFUL_FLOP_PER_PAIR:
            pairs = st["pixel_entry_pairs_blended"]
            out.update(pairs_per_launch=pairs, pairs_per_s=round(pairs / (avg_ms * 1e-3), 1), lane_occupancy=round(st["lane_occupancy"], 4),
                       useful_flop_frac_of_fp32_vector_peak=round(pairs * USEFUL_FLOP_PER_PAIR[kernel] / (avg_ms * 1e-3) / FP32_VECTOR_PEAK, 5))
    except Exception:
        pass
    return out or None


def counter_source():
    """Which committed profile passes `traffic` and `compute` were read from: the manifest names the commit of the
    library sources they were collected on; the line's durations are this run's own."""
    try:
        m = json.load(open(os.path.join(ROOT, "profiles", "counters_manifest.json")))
        return {"collected_on_commit": m.get("collected_on_commit"), "round": m.get("round"), "files": sorted(m.get("files", {})),
                "note": "counters = committed builder-box passes (profiles/counters_manifest.json); durations = this run"}
    except Exception:
        return None


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json, produced by
    tools/collect_hbm_traffic.sh on the same workload: FETCH_SIZE x2, WRITE_SIZE calibrated on the gradient-record memset), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(path)).get(TRAFFIC_NAMES.get(kernel, kernel), {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def reference_gpu_baseline(s, dev, rasterizer, hip_image):
    """The REFERENCE's own rasterizer on this MI355X: oracle/_ref/_ref{2,3}d_C.so is the reference's extension compiled for
    gfx950 from /root/reference by oracle/build_ref.py (baseline leg only, like cpu_baseline; None when it was not built).
    Same scene, same upstream gradients, rasterize_triangles + rasterize_triangles_backward (R2D/ext.cpp:6-8) called
    directly; 1 warm-up + 5 timed steps."""
    import importlib.util
    name = "_ref3d_C" if rasterizer == "3D" else "_ref2d_C"
    path = os.path.join(ROOT, "oracle", "_ref", name + ".so")
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    W, H = s["image_width"], s["image_height"]
    cam = (s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), int(s["sh_degree"]), 1.0, 1.0,
           float(s["background_depth"]), t(s["background"]))
    vertex, shs, opacity, empty = t(s["vertex"]), t(s["shs"]), t(s["opacity"]), torch.empty(0, device=dev)
    g = (t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"]))

    def step():
        out = ref.rasterize_triangles(W, H, *cam, vertex, shs, empty, opacity, False, True, False)
        n, img, radii, depth, normal, csum, cmax, gb, bb, ib = out
        ref.rasterize_triangles_backward(*cam, vertex, shs, empty, opacity, n, radii, gb, bb, ib, *g, True, False)
        return img

    img = step()
    torch.cuda.synchronize()
    steps = 5
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    rel = float((img - hip_image).norm() / img.norm())
    return {"value": round(W * H / (ms * 1e-3) / 1e6, 3), "unit": "Mpix/s", "ms_per_step": round(ms, 3), "kind": "reference",
            "sample": f"the reference's {rasterizer} extension built for gfx950 (oracle/build_ref.py), same scene, {steps} timed steps of "
                      "rasterize_triangles + rasterize_triangles_backward",
            "image_rel_l2_hip_vs_reference": rel}


def cpu_baseline(s, hip_image, variant=2):
    """The CPU oracle (a port of the reference algorithm, OpenMP over tiles, all host cores) on the same scene:
    one fwd+bwd.  The full 1M-triangle / 1080p step is ~15-20 s on 8 cores, which is the bounded sample."""
    from oracle import ts2d_oracle as O
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    import helpers

    cores = O.num_threads()
    t0 = time.perf_counter()
    of = helpers.oracle_forward(s, rich_info=True, variant=variant)
    t1 = time.perf_counter()
    helpers.oracle_backward(s, of, rich_info=True)
    t2 = time.perf_counter()
    W, H = s["image_width"], s["image_height"]
    return {"value": round(W * H / (t2 - t0) / 1e6, 4), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"1 full fwd+bwd step of the same scene ({s['vertex'].shape[0]} triangles, {W}x{H}); "
                      f"fwd {t1 - t0:.2f} s + bwd {t2 - t1:.2f} s on {cores} OpenMP threads",
            "image_rel_l2_hip_vs_oracle": float(helpers.rel_l2(hip_image, of["out_feature"]))}


if __name__ == "__main__":
    main()
