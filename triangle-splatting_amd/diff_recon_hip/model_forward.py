"""Argument construction of the reference model's render call, restated for the HIP rasterizer
(src/diff_recon/models/VanillaTS_model.py:585-694, "VanillaTSModel.forward").

`render_view` takes the model's tensors explicitly (the reference model class, its config system and logger are out of
scope) and performs, in the reference's order:
  * shs = cat(f_dc, f_rest) (:79-80), opacity = sigmoid(raw) (:83-84);
  * gamma rescale of every triangle about its centroid by 1 / sqrt(2^b b Gamma(b)), b = 1 / gamma (:614-618, :445-446);
  * straight-through binarised opacity (:620-621);
  * bg_depth = max |camera_center - vertex| (:623; `background_depth`: one kernel, stays a 0-dim device tensor that the rasterizer
    package converts);
  * render_up_scale: render at s x resolution, bilinear resize of render / depth / normal back, radii // s (:625-659);
  * rich_info = is_training, sh_degree = min(active, max) (:639-641).
"""
from __future__ import annotations

import copy
import ctypes
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from diff_triangle_rasterization_2D import _C as _native

from .losses import downsample_bilinear, downsample_bilinear_many
from .triangle_renderer import TriangleRenderer


def gamma_rescale_ratio(gamma: float) -> float:
    """VanillaTS_model.py:615-617 (scipy.special.gamma == math.gamma for positive reals)."""
    beta = 1.0 / gamma
    return 1.0 / math.sqrt(2.0 ** beta * beta * math.gamma(beta))


def rescale_triangles(vertex: torch.Tensor, ratio) -> torch.Tensor:
    """VanillaTS_model.py:441-447: scale each triangle about its centroid; `ratio` float or (P,) tensor."""
    if isinstance(ratio, torch.Tensor):
        assert ratio.dim() == 1 and ratio.size(0) == vertex.size(0)
        ratio = ratio.unsqueeze(1).unsqueeze(1)
    center = vertex.mean(dim=1, keepdim=True)
    return (vertex - center) * ratio + center


def ste_opacity(opacity: torch.Tensor, threshold: float) -> torch.Tensor:
    """VanillaTS_model.py:621: forward = hard threshold, backward = identity."""
    return ((opacity > threshold).float() - opacity).detach() + opacity


_lib = _native._lib
_lib.tsm_max_vertex_distance.restype = ctypes.c_int
_lib.tsm_max_vertex_distance.argtypes = [ctypes.c_int32] + [ctypes.c_void_p] * 4


def background_depth(vertex: torch.Tensor, camera_center: torch.Tensor) -> torch.Tensor:
    """VanillaTS_model.py:623, `(camera_center - vertex).norm(dim=-1).max()`: a 0-dim device tensor, by one read of the vertices
    (include/ts_model.h: tsm_max_vertex_distance) instead of torch's three kernels.  It is a raster SETTING (the rasterizer package reads it
    as a number), so it carries no gradient here as it carries none into the reference's rasterizer."""
    v = vertex.detach()
    if not v.is_cuda:
        raise RuntimeError("background_depth: the vertices live on the GPU (there is no CPU path)")
    if v.dtype != torch.float32 or not v.is_contiguous():
        v = v.float().contiguous()
    c = camera_center.detach().to(device=v.device, dtype=torch.float32).contiguous()
    out = torch.empty((), device=v.device, dtype=torch.float32)
    _native._check(_lib.tsm_max_vertex_distance(v.numel() // 3, v.data_ptr(), c.data_ptr(), out.data_ptr(),
                                                torch.cuda.current_stream(v.device).cuda_stream), "tsm_max_vertex_distance")
    return out


def render_view(camera, vertex: torch.Tensor, f_dc: torch.Tensor, f_rest: torch.Tensor, raw_opacity: torch.Tensor, *,
                bg_color: torch.Tensor, gamma: float = 1.0, active_sh_degree: int = 0, max_sh_degree: int = 3,
                is_training: bool = True, back_culling: bool = False, gamma_rescale: bool = False,
                ste_threshold: Optional[float] = None, render_up_scale: Optional[int] = None,
                rasterizer_type: str = "3D", shs: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """`shs` (P, M, 3): the colour coefficients as ONE tensor (then f_dc and f_rest are None) -- the layout that saves the reference's
    torch.cat((f_dc, f_rest)) of every forward (VanillaTS_model.py:79-80: 2 x 12 M bytes per triangle moved per step, and again in its
    backward); the reference's two tensors remain the default."""
    if shs is None:
        shs = torch.cat((f_dc, f_rest), dim=1)
    elif f_dc is not None or f_rest is not None:
        raise ValueError("pass either (f_dc, f_rest) or shs")
    opacity = torch.sigmoid(raw_opacity)
    v_render = rescale_triangles(vertex, gamma_rescale_ratio(gamma)) if gamma_rescale else vertex
    o_render = ste_opacity(opacity, ste_threshold) if ste_threshold is not None else opacity
    bg_depth = background_depth(vertex, camera.camera_center)

    w, h = camera.image_width, camera.image_height
    up = int(render_up_scale) if render_up_scale and render_up_scale > 1 else 1
    if up > 1:
        camera = copy.copy(camera)
        camera.image_width, camera.image_height = w * up, h * up

    renderer = TriangleRenderer(camera, bg_depth=bg_depth, bg_color=bg_color, sh_degree=min(active_sh_degree, max_sh_degree),
                                gamma=gamma, back_culling=back_culling, rich_info=is_training, rasterizer_type=rasterizer_type)
    out = renderer.render(v_render, shs, None, o_render)
    if up > 1:
        # F.interpolate(..., size=(h, w), mode="bilinear") of the reference (:649-656); for the integer factor this is, one gather kernel each way
        # (diff_recon_hip.downsample_bilinear, csrc/resample.hip; pinned against F.interpolate + autograd in tests/test_loss_gpu.py)
        # -- and ONE launch for the three images (downsample_bilinear_many: a launch is latency at this size)
        names = [k for k in ("render", "depth", "normal") if k in out]
        for k, small in zip(names, downsample_bilinear_many([out[k] for k in names], (h, w))):
            out[k] = small
        out["radii"] = out["radii"] // up
    pkg = {"render": out["render"]}
    if is_training:  # :664-679
        pkg.update(radii=out["radii"], center2D=out["center2D"], contrib_sum=out["contrib_sum"], contrib_max=out["contrib_max"],
                   depth=out["depth"], normal=out["normal"], opacity=opacity, vertex=vertex, visible_mask=out["radii"] > 0)
    return pkg
