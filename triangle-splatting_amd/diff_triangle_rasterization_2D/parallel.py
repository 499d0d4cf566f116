"""Image-parallel data parallelism for the rasterizer: one process per GPU, every rank holds the full triangle
set and renders a different view, per-triangle gradients are summed over ranks with one RCCL all-reduce.

The reference has no distributed path at all (process-per-scene only,
src/diff_recon/utils/pipeline_utils.py:35-64); this is the new capability BASELINE.json's north_star asks for.
Semantics defined here: gradients of the per-view losses are SUMMED (or averaged with mean=True)
over the views of a step; the densification statistics that the reference model consumes
(src/diff_recon/models/VanillaTS_model.py:347-363) are reduced with the operator the model itself applies
across iterations: radii / contrib_sum / contrib_max with MAX, visibility counts with SUM.

Design for xGMI: the gradient tensors live in ONE contiguous fp32 bucket per step (P*(9+2+1) floats, plus 3M for a dense
SH exchange) that the backward kernels write directly (GradBucket.capture), so RCCL sees a single large message --
on the 8-GPU fully connected xGMI mesh large messages are what reach link bandwidth -- reduced as reduce-scatter +
all-gather on a side stream, so that it overlaps whatever the caller still has queued on the compute stream (the SH
gradient expansion, the loss/backward of a second view).

Works on any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU tensors (used by the
world_size-2 tests that run without a GPU).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """Indices of the views of one step that `rank` renders (round-robin, so any num_views works)."""
    return list(range(rank, num_views, world_size))


class VisibleRows:
    """The triangles that ANY rank saw in this step: the union over the ranks of `radii > 0` (a triangle no view of the step
    sees has an all-zero gradient row on every rank, so its row need not travel).

        rows = VisibleRows(group, device)
        out = raster(...)                       # forward of this rank's view(s)
        rows.begin([out[1], ...])               # radii of every view of this rank: one small MAX all-reduce (P bytes) on a side stream
        ... loss, backward (queued on the compute stream) ...
        bucket.reduce_async(rows=rows)          # gathers the visible rows into a compact buffer, reduces THAT, scatters the sums back
        shx.start(sink, vertex, D, M, rows=rows)

    `begin` is queued right behind the forward and costs the compute stream nothing; `index()` -- called by the consumers AFTER the backward
    has been queued -- waits for the side stream only (the mask has long arrived: it depended on the forward alone) and returns the sorted
    row indices, identical on every rank.  The one host read it needs (the number of visible rows sizes the collectives) therefore happens
    while the compute stream still holds the whole backward.  wire volume = (visible fraction) x the dense exchange; at the headline's
    synthetic scene every triangle lies in the frustum (fraction ~1: nothing saved, two extra 48 MB gathers ~ 40 us), at BASELINE configs[4]'s
    kind of scene (a city, 5 M triangles, a view sees a fraction of it) the exchange shrinks by that fraction."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.device = torch.device(device) if device is not None else None
        self._stream = torch.cuda.Stream(device=device) if (self.device is not None and self.device.type == "cuda") else None
        self._mask = None
        self._idx = None
        self._work = None

    def begin(self, radii_views: Sequence[torch.Tensor]):
        self._idx = None
        dev = radii_views[0].device

        def issue():
            m = radii_views[0] > 0
            for r in radii_views[1:]:
                m = m | (r > 0)
            m = m.to(torch.uint8)
            self._mask = m
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                return dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
            return None

        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._stream):
                self._work = issue()
                for r in radii_views:
                    r.record_stream(self._stream)
        else:
            self._work = issue()

    def index(self) -> torch.Tensor:
        """Sorted indices (int64) of the rows visible on any rank -- the same tensor on every rank.  One host read (the count).  The tensor is
        produced on this object's side stream; the stream that is current when index() is called (the consumer's) is made to wait for it."""
        if self._mask is None:
            raise RuntimeError("VisibleRows.index() before begin()")
        consumer = torch.cuda.current_stream(self._mask.device) if self._stream is not None else None
        if self._idx is None:
            if self._stream is not None:
                with torch.cuda.stream(self._stream):
                    # Work.wait() on RCCL makes the CURRENT stream wait for the collective and nothing else: it must be called with the side
                    # stream current, or nonzero() below -- queued on the side stream -- reads this rank's mask before the MAX has landed and
                    # the ranks size their collectives differently (gloo blocks the host in wait(), which hid it)
                    if self._work is not None:
                        self._work.wait()
                        self._work = None
                    self._idx = torch.nonzero(self._mask).reshape(-1)  # synchronises the SIDE stream only (nonzero reads its count back)
            else:
                if self._work is not None:
                    self._work.wait()
                    self._work = None
                self._idx = torch.nonzero(self._mask).reshape(-1)
        if consumer is not None:
            consumer.wait_stream(self._stream)  # nonzero's second kernel (the indices) may still be in flight on the side stream
            self._idx.record_stream(consumer)
        return self._idx

    @property
    def stream(self):
        return self._stream


class GradBucket:
    """One contiguous fp32 buffer for a fixed list of gradient tensors, summed over the ranks on a side stream.

    * `capture()`: while active, the backward pass of TriangleRasterizer (2D and 3D packages) writes dL_dvertex / dL_dopacity /
      dL_dcenter2D (and a dense dL_dshs / dL_dfeature when the bucket has a slot for it) STRAIGHT into the bucket's views --
      the kernels take the output pointers through the C ABI, so no pack copy exists.  A second backward under the same capture
      (several views per rank) is added to the first.  Under a capture autograd receives NO gradient for the captured slots
      (`param.grad` stays as it was): the gradients come from `wait()`; dL_dcenter2D, the per-view densification statistic, is
      still delivered to each view's own center2D tensor.
    * `reduce_async()` = reduce-scatter + all-gather of the flat buffer: on the fully connected xGMI mesh every
      GPU sends each peer exactly the 1 / world slice that peer owns, over all seven links at once, and gets the reduced slices
      back the same way; between the two halves a rank owns its reduced slice, which is where a sharded optimizer step would go.
      `mode="all_reduce"` keeps the single collective.  Expected exchange times are tabulated in DESIGN.md section 6 (unmeasured:
      no multi-GPU node was available to this project so far)."""

    SLOTS = ("vertex", "opacity", "center2D", "color")  # names a capture can fill; `color` = dL_dshs or dL_dfeature

    def __init__(self, shapes: Sequence[torch.Size], device, dtype=torch.float32, group=None, mean: bool = False,
                 names: Optional[Sequence[str]] = None, mode: str = "rs_ag", force_collectives: bool = False):
        self.shapes = [torch.Size(s) for s in shapes]
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        self.names = list(names) if names is not None else [None] * len(self.shapes)
        self.group = group
        self.mean = mean
        self.mode = mode
        # force_collectives: issue the collectives even in a group of ONE rank -- how the RCCL reduce-scatter + all-gather branch (in-place
        # slices, side-stream ordering) is executed on a single-GPU box (tests/test_multigpu_gpu.py); a no-op arithmetically
        self.force_collectives = force_collectives
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        total = sum(self.numels)
        self.padded = -(-total // (4 * world)) * (4 * world)  # equal 16-byte-aligned slices for reduce-scatter
        self.flat = torch.zeros(self.padded, device=device, dtype=dtype)
        self._stream = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self._work = None
        self._filled = False
        # set by an owner that applies the captured gradients to ITS OWN parameters (diff_recon_hip.ShardedAdam): {slot: leaf tensor}.  A
        # backward under capture() then insists that the rasterizer's inputs for these slots are exactly these leaves.
        self.expected_inputs: Optional[Dict[str, torch.Tensor]] = None

    def views(self) -> List[torch.Tensor]:
        out, off = [], 0
        for shape, n in zip(self.shapes, self.numels):
            out.append(self.flat[off:off + n].view(shape))
            off += n
        return out

    def named_views(self) -> Dict[str, torch.Tensor]:
        return {n: v for n, v in zip(self.names, self.views()) if n is not None}

    def pack(self, grads: Iterable[Optional[torch.Tensor]]):
        for dst, g in zip(self.views(), grads):
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr():
                dst.copy_(g)

    def capture(self):
        """Context manager: rasterizer backward passes inside it write their gradients into this bucket (see class doc)."""
        return _BucketCapture(self)

    def reduce_async(self, rows: Optional["VisibleRows"] = None):
        """Starts the cross-rank sum of the bucket; on GPUs it runs on a side stream ordered after the current stream.  With `rows`
        (VisibleRows of this step) only the rows some rank saw travel: they are gathered into a compact buffer -- tensor after tensor like
        the bucket itself --, that buffer is reduced, and the sums are scattered back; every other row is an exact zero on every rank already."""
        self._filled = False
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(self.group) == 1 and not self.force_collectives):
            return
        world = dist.get_world_size(self.group)

        def reduce_flat(flat, padded):
            # gloo (the CPU / functional-test backend) has no reduce_scatter_tensor: it takes the single all-reduce
            if self.mode == "rs_ag" and dist.get_backend(self.group) != "gloo":
                rank = dist.get_rank(self.group)
                n = padded // world
                mine = flat[rank * n:(rank + 1) * n]
                dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM, group=self.group)
                return dist.all_gather_into_tensor(flat, mine, group=self.group, async_op=True)
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        def issue():
            if rows is None:
                return reduce_flat(self.flat, self.padded)
            idx = rows.index()
            V = int(idx.numel())
            views = self.views()
            P = views[0].shape[0]
            widths = [n // P for n in self.numels]
            total = V * sum(widths)
            padded = -(-max(total, 1) // (4 * world)) * (4 * world)
            compact = torch.zeros(padded, device=self.flat.device, dtype=self.flat.dtype)
            segs, off = [], 0
            for v, w in zip(views, widths):
                seg = compact[off:off + V * w].view(V, w)
                torch.index_select(v.reshape(P, w), 0, idx, out=seg)
                segs.append(seg)
                off += V * w
            work = reduce_flat(compact, padded)
            self._sparse = (work, idx, segs, views, widths, compact)
            self.last_exchanged_bytes = padded * self.flat.element_size()
            return None

        self._sparse = None
        self.last_exchanged_bytes = self.padded * self.flat.element_size()
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            if rows is not None and rows.stream is not None:
                self._stream.wait_stream(rows.stream)
            with torch.cuda.stream(self._stream):
                self._work = issue()
        else:
            self._work = issue()

    # ---- ranged exchange (round 5): the exchange of triangle range k starts while the backward's per-triangle kernel works on range k + 1 ----
    def prepare_ranges(self, num_ranges: int):
        """Call once (K >= 2; every tensor of the bucket must have the P triangles as its first dimension).  Backward passes under capture() then
        run their per-triangle kernel as K launches with an event behind each (include/ts2d.h: ts2d_backward_ranged), and reduce_ranges_async()
        reduces range after range as the events fire.  What this can hide is bounded by the per-triangle kernel itself -- the last ~7 % of a
        backward (0.107 ms at 1 M triangles); the exchange of the LAST range is as exposed as ever."""
        from . import _C
        P = self.shapes[0][0]
        if any(sh[0] != P for sh in self.shapes):
            raise ValueError("a ranged exchange needs every tensor of the bucket to have the triangle count as its first dimension")
        self.num_ranges = int(num_ranges)
        self.range_rows = _C.backward_range_rows(P, self.num_ranges) if self.flat.is_cuda else -(-(-(-P // self.num_ranges)) // 64) * 64
        self.range_events = None
        if self.flat.is_cuda:
            self.range_events = [torch.cuda.Event() for _ in range(self.num_ranges)]
            for e in self.range_events:
                e.record()  # creates the native event (the library records it again behind each range)
        self._ranges_recorded = False

    def reduce_ranges_async(self):
        """The bucket's sum, range by range: all-reduce of rows [k R, (k + 1) R) of every tensor as soon as range k of the backward is done."""
        self._filled = False
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(self.group) == 1 and not self.force_collectives):
            return
        P = self.shapes[0][0]
        views = self.views()
        recorded = getattr(self, "_ranges_recorded", False)

        def issue():
            works = []
            for k in range(self.num_ranges):
                r0, r1 = k * self.range_rows, min(P, (k + 1) * self.range_rows)
                if r0 >= r1:
                    break
                if self._stream is not None and self.range_events and recorded:
                    self._stream.wait_event(self.range_events[k])
                for v in views:
                    works.append(dist.all_reduce(v[r0:r1], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return works

        self.last_exchanged_bytes = self.padded * self.flat.element_size()
        if self._stream is not None:
            if not recorded:  # nobody recorded the events (no backward ran under the capture): order behind the compute stream as a whole
                self._stream.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self._stream):
                self._range_works = issue()
        else:
            self._range_works = issue()
        self._ranges_recorded = False

    all_reduce_async = reduce_async  # round-1 name

    def wait(self) -> List[torch.Tensor]:
        for w in getattr(self, "_range_works", None) or []:
            w.wait()
        if getattr(self, "_range_works", None):
            self._range_works = None
            if self._stream is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)
        sparse = getattr(self, "_sparse", None)
        if sparse is not None:
            # the compact buffer's sums go back to their rows (on the side stream: the collective's wait() orders it behind the exchange)
            work, idx, segs, views, widths, _keep = sparse
            ctx = torch.cuda.stream(self._stream) if self._stream is not None else _null_context()
            with ctx:
                work.wait()
                for v, w, seg in zip(views, widths, segs):
                    v.reshape(v.shape[0], w).index_copy_(0, idx, seg)
            self._sparse = None
            if self._stream is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._stream is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(self._stream)
        if self.mean and dist.is_available() and dist.is_initialized():
            self.flat.div_(dist.get_world_size(self.group))
        return self.views()

    def all_reduce(self, grads: Iterable[Optional[torch.Tensor]]) -> List[torch.Tensor]:
        self.pack(grads)
        self.reduce_async()
        return self.wait()


class _null_context:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


class _BucketCapture:
    def __init__(self, bucket: GradBucket):
        self.bucket = bucket

    def __enter__(self):
        import diff_triangle_rasterization_2D as pkg
        self._prev = pkg._grad_bucket
        pkg._grad_bucket = self.bucket
        self.bucket._filled = False
        return self.bucket

    def __exit__(self, *exc):
        import diff_triangle_rasterization_2D as pkg
        pkg._grad_bucket = self._prev
        return False


def all_reduce_triangle_grads(params: Sequence[torch.Tensor], group=None, mean: bool = False,
                              bucket: Optional[GradBucket] = None) -> GradBucket:
    """Sums `.grad` of the given parameters (vertex, shs/feature, opacity, center2D, ...) over all ranks through
    one flat bucket and writes the reduced values back into the `.grad` tensors."""
    grads = [p.grad for p in params]
    if bucket is None:
        bucket = GradBucket([p.shape for p in params], params[0].device, params[0].dtype, group, mean)
    reduced = bucket.all_reduce(grads)
    for p, r in zip(params, reduced):
        if p.grad is None:
            p.grad = r.clone()
        else:
            p.grad.copy_(r)
    return bucket


def reduce_render_stats(stats: Dict[str, torch.Tensor], group=None) -> Dict[str, torch.Tensor]:
    """Cross-view reduction of the per-triangle statistics returned with rich_info: MAX for radii / contrib_sum /
    contrib_max (the model keeps running maxima of them, VanillaTS_model.py:360-363), SUM for anything named
    '*count*' (visibility counters)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return stats
    out = {}
    for k, v in stats.items():
        v = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM if "count" in k else dist.ReduceOp.MAX, group=group)
        out[k] = v
    return out


# ---- factored SH-gradient exchange ---------------------------------------------------------------------------------
# Per view, dL/dshs of a triangle is the rank-1 product basis_k(dir) * dL_dRGB (R2D/src/backward.cu:9-119): 3 M floats
# that carry 3 floats of information plus the camera centre.  With M = 16 the dense all-reduce bucket is 60 floats per
# triangle (240 MB at 1 M triangles), 48 of them SH; xGMI links are the scarce resource (7 x ~153 GB/s point-to-point
# per GPU), so the SH part travels factored: every rank all-gathers (dL_dRGB, campos) -- 3 floats per triangle and view
# -- and rebuilds sum_v basis(dir_v) x dL_dRGB_v locally with one HBM-bound kernel (csrc/shgrad.hip).  The rest
# (vertex 9 + opacity 1 + center2D 2 floats) stays on the flat all-reduce bucket.  Wire volume per GPU at 8 ranks:
# 2 * 7/8 * 240 MB = 420 MB dense  ->  2 * 7/8 * 48 MB + 7 * 12 MB = 168 MB factored.

class ShGradSink:
    """Collects (dL_dRGB (P,3), campos (3,)) of every SH-mode backward pass run under `factored_sh_grads()`."""

    def __init__(self):
        self.colors: List[torch.Tensor] = []
        self.campos: List[torch.Tensor] = []

    def append(self, dL_dcolor: torch.Tensor, campos: torch.Tensor):
        self.colors.append(dL_dcolor)
        self.campos.append(campos.detach().reshape(3).to(dL_dcolor.device, torch.float32))

    def clear(self):
        self.colors.clear()
        self.campos.clear()


class factored_sh_grads:
    """Context manager: backward passes of TriangleRasterizer (2D and 3D packages) run inside it hand their SH gradients
    to the returned sink in factored form and leave `shs.grad` untouched; `exchange_factored_sh_grads` finishes the job."""

    def __init__(self, sink: Optional[ShGradSink] = None, enabled: bool = True):
        self.sink = sink if sink is not None else ShGradSink()
        self.enabled = enabled

    def __enter__(self) -> ShGradSink:
        import diff_triangle_rasterization_2D as pkg
        self._prev = pkg._sh_grad_sink
        if self.enabled:
            pkg._sh_grad_sink = self.sink
        return self.sink

    def __exit__(self, *exc):
        import diff_triangle_rasterization_2D as pkg
        pkg._sh_grad_sink = self._prev
        return False


def exchange_groups():
    """(bucket_group, sh_group): two process groups over all ranks.  On RCCL every group has its own communicator and stream, so the
    bucket's reduce-scatter + all-gather and the SH-gradient all-gather can be in flight TOGETHER (on one group they would queue behind
    each other); on the ring they use the links in both directions.  Collective call: every rank must call it, in the same order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None, None
    return None, dist.new_group(backend=dist.get_backend())


class FactoredShExchange:
    """`exchange_factored_sh_grads` started on a side stream: `start()` right after the backward, `wait()` where the dense dL_dshs is
    needed (before the optimizer step -- or one step later, see bench.py).  Everything it touches stays referenced until `wait()`."""

    def __init__(self, group=None, device=None):
        self.group = group
        self._stream = torch.cuda.Stream(device=device) if (device is not None and torch.device(device).type == "cuda") else None
        self._out = None
        self._keep = None

    def start(self, sink: ShGradSink, vertex: torch.Tensor, sh_degree: int, M: int, mean: bool = False, expand_fn=None, uniform: bool = True,
              rows: Optional["VisibleRows"] = None):
        if self._stream is None:
            self._out = exchange_factored_sh_grads(sink, vertex, sh_degree, M, self.group, mean, expand_fn, uniform, rows)
            return
        colors, campos = list(sink.colors), list(sink.campos)
        self._keep = (colors, campos, vertex)
        self._stream.wait_stream(torch.cuda.current_stream(vertex.device))
        if rows is not None and rows.stream is not None:
            self._stream.wait_stream(rows.stream)
        with torch.cuda.stream(self._stream):
            self._out = exchange_factored_sh_grads(sink, vertex, sh_degree, M, self.group, mean, expand_fn, uniform, rows)
            for tns in colors + [vertex]:
                tns.record_stream(self._stream)  # allocated on the compute stream, last used on this one

    def wait(self) -> Optional[torch.Tensor]:
        if self._stream is not None and self._out is not None:
            torch.cuda.current_stream(self._out.device).wait_stream(self._stream)
            self._out.record_stream(torch.cuda.current_stream(self._out.device))
        out, self._out, self._keep = self._out, None, None
        return out


def exchange_factored_sh_grads(sink: ShGradSink, vertex: torch.Tensor, sh_degree: int, M: int, group=None,
                               mean: bool = False, expand_fn=None, uniform: bool = False, rows: Optional["VisibleRows"] = None) -> torch.Tensor:
    """All-gathers the sink's factors over the ranks and returns the dense dL_dshs (P, M, 3) summed over every view of
    every rank (divided by the world size with mean=True, like GradBucket).  Ranks may hold different numbers of views
    (the shorter ones are padded with zero-colour rows).  `expand_fn(vertex, campos (V,3), dL_dcolor (V,P,3), sh_degree, M)` defaults to the HIP kernel behind
    `_C.sh_grad_expand`; the CPU tests inject a reference implementation to exercise the protocol over gloo.
    `uniform=True` is the caller's promise that every rank holds the same number of views of the same P triangles (the usual
    training step): the agreement round -- a small all-reduce and a blocking read per step -- is skipped.
    `rows` (VisibleRows of this step): only the colour rows of triangles some rank saw are gathered (the others are exact zeros in every
    view); they are put back into a zero (world * V, P, 3) array in front of the expansion kernel."""
    if not sink.colors:
        raise RuntimeError("no SH-mode backward pass ran under factored_sh_grads()")
    if expand_fn is None:
        from . import _C
        expand_fn = _C.sh_grad_expand
    P = vertex.shape[0]
    V = len(sink.colors)
    # Two messages per rank: the colour factors (V, P, 3) and the camera centres (V, 4: xyz + padding).  They travel separately so that the
    # gathered colours ARE the contiguous (world * V, P, 3) array the expansion kernel reads -- one message with the camera appended to every
    # row made the rows 3 P + 4 floats wide, and slicing the colours out of it was a hidden copy of 12 P V world bytes per step (96 MB at 8
    # ranks and 1 M triangles).
    dev = sink.colors[0].device
    local = torch.empty((V, P, 3), device=dev, dtype=torch.float32)
    local_cam = torch.zeros((V, 4), device=dev, dtype=torch.float32)
    for v, (c, cp) in enumerate(zip(sink.colors, sink.campos)):
        local[v].copy_(c.reshape(P, 3))
        local_cam[v, :3] = cp
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1 and not uniform:
        # ranks may hold different numbers of views (shard_views with num_views % world != 0): agree on the largest count
        # and pad with zero-colour rows, which add nothing to the sum; a different triangle count is a caller error
        meta = torch.tensor([V, -V, P, -P], device=dev, dtype=torch.int64)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
        vmax, pmax, pmin = int(meta[0]), int(meta[2]), -int(meta[3])
        if pmax != pmin:
            raise RuntimeError(f"exchange_factored_sh_grads: ranks disagree on the number of triangles ({pmin} .. {pmax})")
        if vmax != V:
            local = torch.cat([local, torch.zeros((vmax - V, P, 3), device=dev, dtype=torch.float32)], dim=0)
            local_cam = torch.cat([local_cam, torch.zeros((vmax - V, 4), device=dev, dtype=torch.float32)], dim=0)
            V = vmax
    if world > 1 and rows is not None:
        idx = rows.index()
        nvis = int(idx.numel())
        packed = local.index_select(1, idx).contiguous()  # (V, nvis, 3)
        gathered = torch.empty((world * V, nvis, 3), device=dev, dtype=torch.float32)
        cams = torch.empty((world * V, 4), device=dev, dtype=torch.float32)
        dist.all_gather_into_tensor(gathered, packed, group=group)
        dist.all_gather_into_tensor(cams, local_cam, group=group)
        colors = torch.zeros((world * V, P, 3), device=dev, dtype=torch.float32)
        colors.index_copy_(1, idx, gathered)
    elif world > 1:
        colors = torch.empty((world * V, P, 3), device=dev, dtype=torch.float32)
        cams = torch.empty((world * V, 4), device=dev, dtype=torch.float32)
        dist.all_gather_into_tensor(colors, local, group=group)
        dist.all_gather_into_tensor(cams, local_cam, group=group)
    else:
        colors, cams = local, local_cam
    campos = cams[:, :3].contiguous()  # world * V * 12 bytes
    out = expand_fn(vertex.detach(), campos, colors, sh_degree, M)
    if mean and world > 1:
        out.div_(world)
    sink.clear()
    return out
