"""Densification statistics of the reference model, kept by one fused HIP kernel per training iteration
(include/ts_model.h; src/diff_recon/models/VanillaTS_model.py:194-201 state, :347-363 `_training_statistic`,
:228-235 / :309-315 pruning and growth of the state arrays).

`DensificationStats` owns the six per-triangle arrays under the reference's attribute names, so the periodic
densification / pruning rules of the reference (:365-532, eager torch, every few hundred iterations) can read them
unchanged.  `update(render_pkg)` is the per-iteration part; with torch.distributed initialised and `all_views=True` the
per-view inputs are all-gathered first, so every rank applies the statistics of ALL views of the step and the replicas stay
identical.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from diff_triangle_rasterization_2D import _C as _native

_lib = _native._lib
_fp = C.c_void_p
_lib.tsm_training_statistic.restype = C.c_int
_lib.tsm_training_statistic.argtypes = [C.c_int32, C.c_int32] + [_fp] * 11

_STATE = ("gradient_accum", "gradient_denom", "max_radii2D", "contrib_sum", "contrib_max", "contrib_denom")


class DensificationStats:
    def __init__(self, num_triangles: int, device):
        for name in _STATE:  # VanillaTS_model.py:196-201
            setattr(self, name, torch.zeros((num_triangles,), device=device, dtype=torch.float32))

    def __len__(self):
        return self.gradient_accum.shape[0]

    @torch.no_grad()
    def update(self, render_pkg: Dict[str, torch.Tensor], all_views: bool = False, group=None):
        """VanillaTS_model.py:347-363 for the view(s) in `render_pkg` ("radii", "center2D" with .grad populated, and --
        when rendered with rich_info -- "contrib_sum", "contrib_max")."""
        radii = render_pkg["radii"]
        if not radii.is_cuda:
            raise RuntimeError("DensificationStats (MI355X build) needs tensors on a HIP device; there is no CPU fallback")
        grad = render_pkg["center2D"].grad
        if grad is None:
            raise RuntimeError("center2D.grad is not populated: call update() after loss.backward()")
        P = len(self)
        rich = "contrib_sum" in render_pkg
        radii = radii.to(torch.int32).contiguous().view(1, P)
        grad = grad[:, :2].to(torch.float32).contiguous().view(1, P, 2)
        csum = render_pkg["contrib_sum"].contiguous().view(1, P) if rich else None
        cmax = render_pkg["contrib_max"].contiguous().view(1, P) if rich else None
        if all_views and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world = dist.get_world_size(group)

            def gather(t):
                out = torch.empty((world,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
                dist.all_gather_into_tensor(out, t.contiguous(), group=group)
                return out

            radii, grad = gather(radii), gather(grad)
            if rich:
                csum, cmax = gather(csum), gather(cmax)
        V = radii.shape[0]
        with torch.cuda.device(radii.device):
            _native._check(_lib.tsm_training_statistic(
                P, V, radii.data_ptr(), grad.data_ptr(), csum.data_ptr() if rich else None, cmax.data_ptr() if rich else None,
                self.gradient_accum.data_ptr(), self.gradient_denom.data_ptr(), self.max_radii2D.data_ptr(),
                self.contrib_sum.data_ptr(), self.contrib_max.data_ptr(), self.contrib_denom.data_ptr(),
                torch.cuda.current_stream().cuda_stream), "training_statistic")

    def prune(self, prune_mask: torch.Tensor):
        """VanillaTS_model.py:228-234: drop the rows of pruned triangles."""
        keep = ~prune_mask
        for name in _STATE:
            setattr(self, name, getattr(self, name)[keep])

    def grow(self, new_count: int):
        """VanillaTS_model.py:309-315: zero state for appended triangles."""
        for name in _STATE:
            setattr(self, name, F.pad(getattr(self, name), (0, new_count), value=0))
