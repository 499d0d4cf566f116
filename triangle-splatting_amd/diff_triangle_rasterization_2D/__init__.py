"""MI355X-native drop-in for the reference's `diff_triangle_rasterization_2D` Python package.

Public surface -- same names, field order, argument meaning and error behaviour as the reference module
R2D/diff_triangle_rasterization_2D/__init__.py (R2D = submodules/diff-triangle-rasterization-2D):

    TriangleRasterizationSettings   NamedTuple, the reference's 15 fields in the reference's order  (:28-46)
    TriangleRasterizer              nn.Module with `.raster_settings` and
                                    `.forward(vertex, center2D, opacity, shs=None, feature=None)`,
                                    returning 2 or 6 tensors depending on rich_info                 (:167-187)
    _RasterizeTriangles             the torch.autograd.Function behind it                           (:49-164)

so src/diff_recon/renderer/triangle_renderer.py (the direct caller, :3-6, :38-57, :69-75) works unchanged.
The native side is libts2d.so (hand-written HIP for gfx950, C ABI in include/ts2d.h) reached through `_C`;
nothing here falls back to CPU or eager torch.

Deliberate deviations from the reference module:
  * FIX: the reference's backward raises NameError with rich_info=False because
    grad_out_depth / grad_out_normal are never bound (:114-117, :141).  Here they are empty tensors.
  * radii / contrib_sum / contrib_max are marked non-differentiable, and upstream gradients are made
    contiguous instead of tripping the "input tensors must be contiguous" check.
"""
from __future__ import annotations

import contextlib
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class TriangleRasterizationSettings(NamedTuple):
    image_width: int
    image_height: int
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    campos: torch.Tensor
    sh_degree: int
    gamma: float
    scale_modifier: float
    background_depth: float
    background: torch.Tensor
    back_culling: bool
    rich_info: bool
    debug: bool


@contextlib.contextmanager
def _snapshot_on_error(tag: str, args: tuple, enabled: bool):
    """settings.debug behaviour of the reference (:14-25): if the native call throws, a CPU copy of its
    arguments is saved to snapshot_<tag>.dump for offline reproduction, then the error propagates."""
    if not enabled:
        yield
        return
    frozen = tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)
    try:
        yield
    except Exception:
        torch.save(frozen, f"snapshot_{tag}.dump")
        print(f"\nAn error occured in {tag}. Writing snapshot_{tag}.dump for debugging.")
        raise


def _camera_and_geometry_args(rs: TriangleRasterizationSettings, background_depth) -> tuple:
    """The ten settings-derived arguments that both native entry points share, in their positional order
    (tan_fovx .. background; R2D/src/extension_interface.h:7-62)."""
    return (
        rs.tanfovx, rs.tanfovy,
        rs.viewmatrix.contiguous(), rs.projmatrix.contiguous(), rs.campos.contiguous(),
        rs.sh_degree, rs.gamma, rs.scale_modifier, background_depth, rs.background.contiguous(),
    )


# Set by parallel.factored_sh_grads(): while a sink is installed, backward passes in SH mode do not form the dense
# dL_dshs; they append (dL_dRGB (P,3), campos (3,)) to the sink and return no gradient for `shs` (parallel.py rebuilds the
# sum over all ranks' views from the exchanged factors).
_sh_grad_sink = None
# Sync-free forward (include/ts2d.h: ts2d_forward): None = off (the reference's sequence, one blocking read of num_rendered per
# forward); an int, or a callable (P, width, height) -> int, = the capacity in tile instances the binning state is sized for.
# With it on, forward() never waits for the GPU; call `forward_overflowed(out_feature)` (one blocking read) once per step, e.g. after
# the optimizer step has been queued, and re-render with a larger capacity when it reports True (the overflowing forward rendered
# the background only).
_instance_capacity = None


def set_instance_capacity(capacity):
    """Enables (int or callable (P, width, height) -> int) or disables (None) the sync-free forward for every later forward()."""
    global _instance_capacity
    _instance_capacity = capacity


_last_sync_free_forward = None  # (P, width, height, geometryBuffer, imageBuffer) of the most recent sync-free forward


def forward_overflowed(out_feature: torch.Tensor = None):
    """(overflowed, true instance count) of the most recent sync-free forward -- or, given one of its outputs whose graph is still
    alive, of the forward that produced `out_feature`.  One blocking read."""
    if out_feature is not None and out_feature.grad_fn is not None:
        try:
            node = out_feature.grad_fn
            saved = node.saved_tensors
            rs = node.raster_settings
            return _C.forward_status(saved[0].shape[0], rs.image_width, rs.image_height, saved[5], saved[7])
        except RuntimeError:  # the graph was freed by backward(): fall back to the most recent forward
            pass
    if _last_sync_free_forward is None:
        raise RuntimeError("no sync-free forward has run (set_instance_capacity)")
    return _C.forward_status(*_last_sync_free_forward)

_center2D_zeros = {}


def center2D_sink(num_points: int, device, dtype=torch.float32) -> torch.Tensor:
    """The `center2D` argument of TriangleRasterizer.forward: a (P, 2) leaf that exists only so that autograd has somewhere to put
    dL_dcenter2D (`center2D.grad`, read by the model's densification statistics, VanillaTS_model.py:347-363); it is never sent to the native
    side (reference __init__.py:52-60).  The reference's caller makes a fresh `torch.zeros((P, 2), requires_grad=True)` per render call
    (src/diff_recon/renderer/triangle_renderer.py:67): one fill kernel on the critical path of every step for values nobody reads.  This hands
    out a NEW leaf per call -- its own `.grad`, its own place in the graph -- over ONE cached block of zeros per (device, dtype, P): no kernel.
    Do not write into it in place (nothing in the reference does)."""
    key = (torch.device(device), dtype, int(num_points))
    z = _center2D_zeros.get(key)
    if z is None:
        if len(_center2D_zeros) >= 8:  # a model that densifies walks through sizes: keep the cache from growing without bound
            _center2D_zeros.clear()
        z = _center2D_zeros[key] = torch.zeros((int(num_points), 2), device=key[0], dtype=dtype)
    return z.detach().requires_grad_(True)


# Set by parallel.GradBucket.capture(): while a bucket is installed, backward passes write dL_dvertex / dL_dopacity /
# dL_dcenter2D (and the dense colour gradient when the bucket has a slot for it) straight into the bucket's views.
_grad_bucket = None


class _RasterizeTriangles(torch.autograd.Function):
    """autograd inputs: (vertex, center2D, shs, feature, opacity, raster_settings); center2D is a gradient
    sink only and is never sent to the native side (reference :52-60, :156-164)."""

    _variant = 2  # the sibling package diff_triangle_rasterization_3D subclasses this with _variant = 3
                  # (ctx is an instance of the generated backward node; ctx._forward_cls is the class .apply ran on)

    @staticmethod
    def forward(ctx, vertex, center2D, shs, feature, opacity, raster_settings):
        rs = raster_settings
        # background_depth may arrive as a 0-dim device tensor (src/diff_recon/models/VanillaTS_model.py:623: max |campos - vertex|,
        # computed on the device every step).  pybind converts it to float in the reference -- a full device synchronisation per forward
        #  -- here a float32 tensor on the rasterizer's device goes to the kernels as a pointer and nothing waits.
        bg_depth = rs.background_depth
        if isinstance(bg_depth, torch.Tensor):
            if bg_depth.is_cuda and bg_depth.device == vertex.device and bg_depth.numel() == 1:
                bg_depth = bg_depth.detach().to(torch.float32).reshape(1).contiguous()  # no host round trip; same fp32 value
            else:
                bg_depth = float(bg_depth)
        else:
            bg_depth = float(bg_depth)
        native_args = (rs.image_width, rs.image_height) + _camera_and_geometry_args(rs, bg_depth) + (
            vertex, shs, feature, opacity, rs.back_culling, rs.rich_info, rs.debug)
        with _snapshot_on_error("rasterize_triangles", native_args, rs.debug):
            (num_rendered, out_feature, radii, depth, normal, contrib_sum, contrib_max,
             geometryBuffer, binningBuffer, imageBuffer) = _C.rasterize_triangles(
                *native_args, variant=ctx._forward_cls._variant,
                instance_capacity=(_instance_capacity(vertex.shape[0], rs.image_width, rs.image_height) if callable(_instance_capacity)
                                   else _instance_capacity) if vertex.shape[0] > 0 else None)

        if _instance_capacity is not None and vertex.shape[0] > 0:
            global _last_sync_free_forward
            _last_sync_free_forward = (vertex.shape[0], rs.image_width, rs.image_height, geometryBuffer, imageBuffer)
        # autograd would otherwise materialise a zero tensor for every output without an incoming gradient (radii, contrib_sum,
        # contrib_max: three fill kernels per step that nothing reads); backward() makes its own zeros where it needs them
        ctx.set_materialize_grads(False)
        # which differentiable inputs are outputs of autograd operations rather than leaf parameters (sigmoid(raw_opacity), cat(f_dc,
        # f_rest), rescaled vertices: diff_recon_hip/model_forward.py): under GradBucket.capture() those still need their gradient
        # handed to autograd, or the chain rule stops at the rasterizer
        ctx.input_is_leaf = tuple(t.grad_fn is None for t in (vertex, shs, feature, opacity))
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.bg_depth = bg_depth
        ctx.num_channels = int(out_feature.shape[0])
        ctx.save_for_backward(vertex, shs, feature, opacity, radii, geometryBuffer, binningBuffer, imageBuffer)
        if rs.rich_info:
            ctx.mark_non_differentiable(radii, contrib_sum, contrib_max)
            return out_feature, radii, depth, normal, contrib_sum, contrib_max
        ctx.mark_non_differentiable(radii)
        return out_feature, radii

    @staticmethod
    def backward(ctx, *grads_out):
        rs = ctx.raster_settings
        vertex, shs, feature, opacity, radii, geometryBuffer, binningBuffer, imageBuffer = ctx.saved_tensors
        H, W = rs.image_height, rs.image_width
        zeros = lambda *shape: torch.zeros(shape, device=vertex.device, dtype=vertex.dtype)
        g_feature = grads_out[0] if grads_out[0] is not None else zeros(ctx.num_channels, H, W)
        if rs.rich_info and (grads_out[2] is not None or grads_out[3] is not None):
            g_depth = grads_out[2] if grads_out[2] is not None else zeros(H, W)
            g_normal = grads_out[3] if grads_out[3] is not None else zeros(3, H, W)
        else:
            # rich_info False: FIX of the reference's unbound names.  rich_info True and a loss that reads the colours only: empty = "no gradient
            # on depth and normal" (include/ts2d.h, ts2d_loss_grads) -- the values two images of zeros give, without the two fills and by the
            # colour-only pixel kernel
            g_depth = g_normal = torch.empty((0,), device=vertex.device, dtype=vertex.dtype)
        native_args = _camera_and_geometry_args(rs, ctx.bg_depth) + (
            vertex, shs, feature, opacity, ctx.num_rendered, radii, geometryBuffer, binningBuffer, imageBuffer,
            g_feature.contiguous(), g_depth.contiguous(), g_normal.contiguous(), rs.rich_info, rs.debug)
        with _snapshot_on_error("rasterize_triangles_backward", native_args, rs.debug):
            sink = _sh_grad_sink if (ctx.needs_input_grad[2] and shs.numel() > 0) else None
            bucket = _grad_bucket
            if bucket is not None and bucket.expected_inputs is not None:
                # the bucket's owner steps ITS parameters with what lands in the bucket (ShardedAdam): the gradient with respect to the
                # rasterizer's input is the parameter's gradient only if the input IS that parameter
                given = {"vertex": vertex, "opacity": opacity, "color": shs if shs.numel() > 0 else feature}
                for slot, want in bucket.expected_inputs.items():
                    got = given.get(slot)
                    if got is None:
                        continue
                    is_leaf = ctx.input_is_leaf[{"vertex": 0, "opacity": 3, "color": 1 if shs.numel() > 0 else 2}[slot]]
                    if not is_leaf or got.data_ptr() != want.data_ptr() or got.shape != want.shape:
                        raise RuntimeError(f"GradBucket.capture(): the rasterizer's `{slot}` input is not the parameter the bucket's optimizer owns "
                                           "(pass opt.params[...] itself; an operation between the parameter and the rasterizer would make the "
                                           "bucket hold the wrong gradient)")
            place = bucket.named_views() if (bucket is not None and not bucket._filled) else None
            # a bucket prepared for a ranged exchange (GradBucket.prepare_ranges): the per-triangle kernel runs range by range with an event behind
            # each, so that the exchange of range k overlaps range k + 1 -- only for the backward that WRITES the bucket (the first under a capture)
            ranged = place is not None and getattr(bucket, "range_events", None)
            g_vertex, g_center2D, g_shs, g_feat, g_opacity = _C.rasterize_triangles_backward(
                *native_args, variant=ctx._forward_cls._variant, sh_factored=sink is not None, out=place, range_events=ranged or None)
            if ranged:
                bucket._ranges_recorded = True
            if sink is not None:
                sink.append(g_feat, rs.campos)
            if bucket is not None:
                nv = bucket.named_views()
                use_shs = shs.numel() > 0
                captured = (("vertex", g_vertex), ("opacity", g_opacity), ("center2D", g_center2D),
                            ("color", (g_shs if use_shs else g_feat) if sink is None else None))
                if place is None:  # a further view under the same capture: added to what the first one wrote
                    for name, g in captured:
                        if name in nv and g is not None:
                            nv[name].add_(g.view(nv[name].shape))
                    # the range events of the FIRST backward no longer say "rows final": these add_ kernels follow them on the compute stream.
                    # reduce_ranges_async() then orders the whole exchange behind the compute stream instead of range by range
                    if getattr(bucket, "range_events", None):
                        bucket._ranges_recorded = False
                # What autograd gets back under a capture: NOTHING for the captured parameter slots -- their gradient lives in the bucket
                # until bucket.wait() (handing out the bucket's own views would let AccumulateGrad alias `param.grad` to the bucket, and a
                # second view would then be added twice: once above, once by autograd) -- and a PRIVATE tensor for dL_dcenter2D, which
                # is a per-view statistic (each render call owns its center2D, VanillaTS_model.py:347-363), not a parameter gradient.
                # That holds for LEAF parameters only.  An input that is itself the output of autograd operations (sigmoid(raw_opacity), a
                # rescaled vertex tensor, cat(f_dc, f_rest)) gets a private copy, so that the operations upstream of the rasterizer still
                # differentiate; the bucket then holds the gradient with respect to the rasterizer's input, which is what it is asked for.
                leaf_vertex, leaf_shs, leaf_feature, leaf_opacity = ctx.input_is_leaf
                private = (lambda g: g.clone()) if place is not None else (lambda g: g)  # a later view's tensors are not the bucket's
                if "vertex" in nv:
                    g_vertex = None if leaf_vertex else private(g_vertex)
                if "opacity" in nv:
                    g_opacity = None if leaf_opacity else private(g_opacity)
                if "color" in nv and sink is None:
                    if use_shs:
                        g_shs = None if leaf_shs else private(g_shs)
                    else:
                        g_feat = None if leaf_feature else private(g_feat)
                if "center2D" in nv and place is not None:
                    g_center2D = g_center2D.clone()
                bucket._filled = True
        # The placeholder standing in for the unused one of shs/feature is a CPU `torch.Tensor([])`
        # (reference :183-184) that never requires grad; hand autograd None for it.
        if not ctx.needs_input_grad[2]:
            g_shs = None
        if not ctx.needs_input_grad[3]:
            g_feat = None
        return g_vertex, g_center2D, g_shs, g_feat, g_opacity, None


class TriangleRasterizer(nn.Module):
    _function = _RasterizeTriangles

    def __init__(self, raster_settings: TriangleRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, vertex, center2D, opacity, shs=None, feature=None):
        if (shs is None) == (feature is None):
            # same exception type and message as the reference (:180-181)
            raise Exception("Please provide excatly one of either SHs or feature!")
        placeholder = torch.Tensor([])  # stays on the CPU like the reference's; never dereferenced
        return self._function.apply(
            vertex, center2D, placeholder if shs is None else shs, placeholder if feature is None else feature,
            opacity, self.raster_settings)


set_capacity_hint_key = _C.set_capacity_hint_key      # separate binning-size histories per camera group / model (include/ts2d.h)
speculative_overflows = _C.speculative_overflows      # forwards that guessed too small and rendered twice, since the process started

__all__ = ["TriangleRasterizationSettings", "TriangleRasterizer", "set_instance_capacity", "forward_overflowed", "set_capacity_hint_key",
           "speculative_overflows", "center2D_sink"]
