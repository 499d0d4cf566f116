"""Photometric losses backed by the fused HIP kernels of libts2d.so (include/ts_loss.h).

Same call surface as the reference's `L1`, `SSIMLoss`, `ssimLoss` (src/diff_recon/trainers/trainer_utils.py:323-324, 96-103,
349): `L1(t1, t2)` is differentiable with respect to both arguments (the trainer's affine regulariser passes two renders,
VanillaTS_trainer.py:103); `ssimLoss(img1, img2)` and the fused loss with respect to their FIRST argument (the render; the
ground truth never requires grad in the trainers) -- a second argument that requires grad raises instead of dropping it.  Inputs of 2, 3 or 4 dimensions are accepted
like `normalize_shape` (trainer_utils.py:80-93); a batch dimension folds into channels, which is what the reference's
depthwise convolution + global mean computes.

`PhotometricLoss(w_L1, w_ssim)(image, gt)` / `photometric_loss(image, gt, w_L1, w_ssim)` evaluate
    w_L1 * L1(image, gt) + w_ssim * ssimLoss(image, gt)          (VanillaTS_trainer.py:80-81,111)
in ONE forward launch (+ a one-block finisher) and ONE backward launch; the reference spends ~20 eager kernels on it.
The maintainer-side change is three lines in VanillaTS_trainer.py.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from diff_triangle_rasterization_2D import _C as _native

_lib = _native._lib
_fp = C.c_void_p
_lib.tsl_workspace_bytes.restype = C.c_size_t
_lib.tsl_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
_lib.tsl_photometric_forward.restype = C.c_int
_lib.tsl_photometric_forward.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, _fp,
                                         C.c_size_t, _fp, _fp]
_lib.tsl_photometric_backward.restype = C.c_int
_lib.tsl_photometric_backward.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _fp, C.c_size_t,
                                          _fp, _fp, _fp]


def _chw(img1: torch.Tensor, img2: torch.Tensor):
    """normalize_shape of the reference (trainer_utils.py:80-93), folded to (C, H, W)."""
    if img1.size() != img2.size():
        raise ValueError("Input images must have the same dimensions.")
    if img1.dim() == 4:
        c = img1.size(0) * img1.size(1)
    elif img1.dim() == 3:
        c = img1.size(0)
    elif img1.dim() == 2:
        c = 1
    else:
        raise ValueError("Input images must have 2, 3, or 4 dimensions.")
    return c, img1.size(-2), img1.size(-1)


def _check_inputs(image: torch.Tensor, gt: torch.Tensor):
    if not image.is_cuda or not gt.is_cuda:
        raise RuntimeError("the photometric loss (MI355X build) needs tensors on a HIP device; there is no CPU fallback")
    if image.dtype != torch.float32 or gt.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float")


class _Photometric(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, w_l1, w_ssim):
        c, h, w = _chw(image, gt)
        _check_inputs(image, gt)
        image_c, gt_c = image.contiguous(), gt.contiguous()
        dev = image.device
        # the SSIM term is differentiated with respect to its FIRST argument only (as every caller in the reference uses it); a
        # second argument that requires grad there would silently lose its gradient, so it is refused.  The L1 term is
        # antisymmetric and hands -g to the second argument (VanillaTS_trainer.py:103 passes two renders to L1).
        if gt.requires_grad and float(w_ssim) != 0.0:
            raise RuntimeError("ssimLoss / photometric_loss (MI355X build) differentiate with respect to the first argument only; "
                               "detach the second one or swap the arguments")
        need_grad = bool(image.requires_grad or gt.requires_grad)
        with torch.cuda.device(dev):
            nbytes = _lib.tsl_workspace_bytes(c, h, w)
            ws = torch.empty((nbytes,), device=dev, dtype=torch.uint8)
            out = torch.empty((3,), device=dev, dtype=torch.float32)
            _native._check(_lib.tsl_photometric_forward(image_c.data_ptr(), gt_c.data_ptr(), c, h, w, float(w_l1), float(w_ssim),
                                                        int(need_grad), ws.data_ptr(), nbytes, out.data_ptr(),
                                                        torch.cuda.current_stream().cuda_stream), "photometric_loss")
        ctx.dims = (c, h, w)
        ctx.weights = (float(w_l1), float(w_ssim))
        ctx.save_for_backward(image_c, gt_c, ws)
        ctx.parts = out  # out[1] = L1, out[2] = 1 - SSIM (detached diagnostics)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        image, gt, ws = ctx.saved_tensors
        c, h, w = ctx.dims
        w_l1, w_ssim = ctx.weights
        with torch.cuda.device(image.device):
            g = torch.empty_like(image)
            go = grad_out.contiguous().to(torch.float32)
            _native._check(_lib.tsl_photometric_backward(image.data_ptr(), gt.data_ptr(), c, h, w, w_l1, w_ssim, ws.data_ptr(),
                                                         ws.numel(), go.data_ptr(), g.data_ptr(),
                                                         torch.cuda.current_stream().cuda_stream), "photometric_loss backward")
        return (g if ctx.needs_input_grad[0] else None), (-g if ctx.needs_input_grad[1] else None), None, None


def photometric_loss(image: torch.Tensor, gt: torch.Tensor, w_L1: float, w_ssim: float) -> torch.Tensor:
    """w_L1 * mean|image - gt| + w_ssim * (1 - SSIM(image, gt)), fused (VanillaTS_trainer.py:80-81,111)."""
    return _Photometric.apply(image, gt, w_L1, w_ssim)


def L1(t1: torch.Tensor, t2: torch.Tensor) -> torch.Tensor:
    """trainer_utils.py:323-324 for image-shaped inputs (2-4 dims)."""
    return _Photometric.apply(t1, t2, 1.0, 0.0)


class SSIMLoss(nn.Module):
    """trainer_utils.py:96-103: 1 - SSIM with the 11x11, sigma 1.5 Gaussian window and zero padding."""

    def forward(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        return _Photometric.apply(img1, img2, 0.0, 1.0)


ssimLoss = SSIMLoss()  # trainer_utils.py:349


class PhotometricLoss(nn.Module):
    def __init__(self, w_L1: float, w_ssim: float):
        super().__init__()
        self.w_L1, self.w_ssim = float(w_L1), float(w_ssim)

    def forward(self, image: torch.Tensor, gt_image: torch.Tensor) -> torch.Tensor:
        return _Photometric.apply(image, gt_image, self.w_L1, self.w_ssim)


# ---- depth / normal consistency loss (trainer_utils.py:204-257) -------------------------------------------------------------------
_lib.tsl_depth_normal_workspace_bytes.restype = C.c_size_t
_lib.tsl_depth_normal_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_double]
_lib.tsl_depth_normal_forward.restype = C.c_int
_lib.tsl_depth_normal_forward.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_double, C.c_float, _fp, C.c_size_t, _fp, _fp]
_lib.tsl_depth_normal_backward.restype = C.c_int
_lib.tsl_depth_normal_backward.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_double, _fp, C.c_size_t, _fp, _fp, _fp, _fp]


class _DepthNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, normal, tan_fovx, tan_fovy, scale_factor, quantile):
        if not depth.is_cuda or not normal.is_cuda:
            raise RuntimeError("DepthNormalLoss (MI355X build) needs tensors on a HIP device; there is no CPU fallback")
        if depth.dtype != torch.float32 or normal.dtype != torch.float32:
            raise RuntimeError("expected scalar type Float")
        if depth.dim() != 2 or normal.dim() != 3 or normal.shape[0] != 3 or tuple(normal.shape[1:]) != tuple(depth.shape):
            raise ValueError("expected depth (H, W) and normal (3, H, W)")
        H, W = depth.shape
        d, n = depth.contiguous(), normal.contiguous()
        scale = float(scale_factor) if scale_factor is not None else 1.0
        with torch.cuda.device(depth.device):
            nbytes = _lib.tsl_depth_normal_workspace_bytes(H, W, scale)
            ws = torch.empty((nbytes,), device=depth.device, dtype=torch.uint8)
            out = torch.empty((1,), device=depth.device, dtype=torch.float32)
            _native._check(_lib.tsl_depth_normal_forward(d.data_ptr(), n.data_ptr(), H, W, float(tan_fovx), float(tan_fovy), scale, float(quantile),
                                                         ws.data_ptr(), nbytes, out.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           "depth_normal_loss")
        ctx.args = (H, W, float(tan_fovx), float(tan_fovy), scale)
        ctx.save_for_backward(d, n, ws)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        d, n, ws = ctx.saved_tensors
        H, W, tx, ty, scale = ctx.args
        with torch.cuda.device(d.device):
            gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
            gn = torch.empty_like(n) if ctx.needs_input_grad[1] else None
            go = grad_out.contiguous().to(torch.float32)
            _native._check(_lib.tsl_depth_normal_backward(d.data_ptr(), n.data_ptr(), H, W, tx, ty, scale, ws.data_ptr(), ws.numel(), go.data_ptr(),
                                                          gd.data_ptr() if gd is not None else None, gn.data_ptr() if gn is not None else None,
                                                          torch.cuda.current_stream().cuda_stream), "depth_normal_loss backward")
        return gd, gn, None, None, None, None


class DepthNormalLoss(nn.Module):
    """trainer_utils.py:204-257 with the reference's constructor and call surface:
    `DepthNormalLoss(scale_factor=0.5)(depth, normal, cam.tan_fovx, cam.tan_fovy)` (VanillaTS_trainer.py:30-31,84).  One fused forward
    (three elementwise launches + the library's radix sort for the quantile) and four launches backward instead of ~100 eager kernels."""

    def __init__(self, depth_grad: bool = True, normal_grad: bool = True, scale_factor: float = None, depth_grad_filter_quantile: float = 0.9):
        super().__init__()
        self.depth_grad = depth_grad
        self.normal_grad = normal_grad
        self.scale_factor = scale_factor
        self.depth_grad_filter_quantile = depth_grad_filter_quantile

    def forward(self, depth: torch.Tensor, normal: torch.Tensor, tan_fovx: float, tan_fovy: float) -> torch.Tensor:
        if not self.depth_grad:
            depth = depth.detach()
        if not self.normal_grad:
            normal = normal.detach()
        return _DepthNormal.apply(depth, normal, tan_fovx, tan_fovy, self.scale_factor, self.depth_grad_filter_quantile)


# ---- DoGLoss / SmoothnessLoss (round 5; csrc/aux_losses.hip) -------------------------------------------------------------------------
_lib.tsl_aux_loss_workspace_bytes.restype = C.c_size_t
_lib.tsl_aux_loss_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double]
_lib.tsl_dog_mask.restype = C.c_int
_lib.tsl_dog_mask.argtypes = [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_int32, C.c_int32, C.c_double, _fp, C.c_size_t, _fp, _fp]
_lib.tsl_smoothness_mask.restype = C.c_int
_lib.tsl_smoothness_mask.argtypes = [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_float, _fp, C.c_size_t, _fp, _fp]
_lib.tsl_masked_l1_forward.restype = C.c_int
_lib.tsl_masked_l1_forward.argtypes = [_fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_size_t, _fp, _fp]
_lib.tsl_masked_l1_backward.restype = C.c_int
_lib.tsl_masked_l1_backward.argtypes = [_fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp]
_lib.tsl_scharr_smoothness_forward.restype = C.c_int
_lib.tsl_scharr_smoothness_forward.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_size_t, _fp, _fp]
_lib.tsl_scharr_smoothness_backward.restype = C.c_int
_lib.tsl_scharr_smoothness_backward.argtypes = [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_size_t, _fp, _fp, _fp]


def _aux_prepare(img: torch.Tensor, img_gt: torch.Tensor, what: str):
    _check_inputs(img, img_gt)
    c, h, w = _chw(img, img_gt)
    if c > 8:
        raise RuntimeError(f"{what} (MI355X build) takes at most 8 channels (batch x channels)")
    if img_gt.requires_grad:
        raise RuntimeError(f"{what}: the target image must not require grad (its mask is formed without gradient, as in the reference)")
    return c, h, w, img.contiguous(), img_gt.contiguous()


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, mask, c, h, w, ws):
        out = torch.empty((1,), device=img.device, dtype=torch.float32)
        with torch.cuda.device(img.device):
            _native._check(_lib.tsl_masked_l1_forward(img.data_ptr(), gt.data_ptr(), mask.data_ptr(), c, h, w, ws.data_ptr(), ws.numel(), out.data_ptr(),
                                                      torch.cuda.current_stream().cuda_stream), "DoGLoss")
        ctx.shape = (c, h, w)
        ctx.save_for_backward(img, gt, mask)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        img, gt, mask = ctx.saved_tensors
        c, h, w = ctx.shape
        g = torch.empty_like(img)
        go = grad_out.contiguous().to(torch.float32)
        with torch.cuda.device(img.device):
            _native._check(_lib.tsl_masked_l1_backward(img.data_ptr(), gt.data_ptr(), mask.data_ptr(), c, h, w, go.data_ptr(), g.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream), "DoGLoss backward")
        return g, None, None, None, None, None, None


class _ScharrSmoothness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, mask, c, h, w, ws):
        out = torch.empty((1,), device=img.device, dtype=torch.float32)
        with torch.cuda.device(img.device):
            _native._check(_lib.tsl_scharr_smoothness_forward(img.data_ptr(), mask.data_ptr(), c, h, w, ws.data_ptr(), ws.numel(), out.data_ptr(),
                                                              torch.cuda.current_stream().cuda_stream), "SmoothnessLoss")
        ctx.shape = (c, h, w)
        ctx.save_for_backward(img, mask, ws)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        img, mask, ws = ctx.saved_tensors
        c, h, w = ctx.shape
        g = torch.empty_like(img)
        go = grad_out.contiguous().to(torch.float32)
        with torch.cuda.device(img.device):
            _native._check(_lib.tsl_scharr_smoothness_backward(img.data_ptr(), mask.data_ptr(), c, h, w, ws.data_ptr(), ws.numel(), go.data_ptr(), g.data_ptr(),
                                                               torch.cuda.current_stream().cuda_stream), "SmoothnessLoss backward")
        return g, None, None, None, None, None


class DoGLoss(nn.Module):
    """trainer_utils.py:124-148 with the reference's constructor and call surface: `DoGLoss(freq=90, scale_factor=0.5)(img, img_gt)` =
    L1(img * mask, img_gt * mask), mask = the thresholded, normalised difference of Gaussians of the down-sampled grey target (no gradient).
    `mask(img_gt)` returns the (H, W) mask alone (it depends on the target only: a caller that keeps its targets can keep their masks)."""

    def __init__(self, freq: int = 90, scale_factor: float = 0.5):
        super().__init__()
        self.freq = freq
        self.scale_factor = scale_factor
        sigma = 0.1 + (100 - freq) * 0.1 if freq >= 50 else 0.1 + freq * 0.1  # DoGFilter's argument, trainer_utils.py:129
        self.sigma1, self.sigma2 = sigma, 2 * sigma                          # :108-109
        self.kernel_size1 = int(2 * round(3 * self.sigma1) + 1)              # :110-111 (Python's round)
        self.kernel_size2 = int(2 * round(3 * self.sigma2) + 1)
        if self.kernel_size2 > 33:
            raise ValueError("DoGLoss (MI355X build): kernel sizes up to 33 (sigma <= 2.6, i.e. freq >= 75 or freq <= 25)")

    def _workspace(self, ref: torch.Tensor, c: int, h: int, w: int):
        scale = float(self.scale_factor) if self.scale_factor is not None else 1.0
        with torch.cuda.device(ref.device):
            return torch.empty((_lib.tsl_aux_loss_workspace_bytes(c, h, w, scale),), device=ref.device, dtype=torch.uint8), scale

    @torch.no_grad()
    def mask(self, img_gt: torch.Tensor, _ws=None) -> torch.Tensor:
        c, h, w, gt, _ = _aux_prepare(img_gt, img_gt.detach(), "DoGLoss")
        ws, scale = _ws if _ws is not None else self._workspace(gt, c, h, w)
        m = torch.empty((h, w), device=gt.device, dtype=torch.float32)
        with torch.cuda.device(gt.device):
            _native._check(_lib.tsl_dog_mask(gt.data_ptr(), c, h, w, self.sigma1, self.kernel_size1, self.sigma2, self.kernel_size2, int(self.freq >= 50), scale,
                                             ws.data_ptr(), ws.numel(), m.data_ptr(), torch.cuda.current_stream().cuda_stream), "DoGLoss mask")
        return m

    def forward(self, img: torch.Tensor, img_gt: torch.Tensor) -> torch.Tensor:
        c, h, w, x, gt = _aux_prepare(img, img_gt, "DoGLoss")
        ws = self._workspace(gt, c, h, w)
        return _MaskedL1.apply(x, gt, self.mask(gt, ws), c, h, w, ws[0])


class SmoothnessLoss(nn.Module):
    """trainer_utils.py:181-201: `SmoothnessLoss(quantile=0.3, scale_factor=0.5)(img, img_gt)` = mean(|Scharr(img)|_2 * mask), mask = where the
    up-sampled gradient norm of the down-sampled target lies below its `quantile` (no gradient)."""

    def __init__(self, quantile: float = 0.3, scale_factor: float = 0.5):
        super().__init__()
        self.quantile = quantile
        self.scale_factor = scale_factor

    def _workspace(self, ref: torch.Tensor, c: int, h: int, w: int):
        scale = float(self.scale_factor) if self.scale_factor is not None else 1.0
        with torch.cuda.device(ref.device):
            return torch.empty((_lib.tsl_aux_loss_workspace_bytes(c, h, w, scale),), device=ref.device, dtype=torch.uint8), scale

    @torch.no_grad()
    def mask(self, img_gt: torch.Tensor, _ws=None) -> torch.Tensor:
        c, h, w, gt, _ = _aux_prepare(img_gt, img_gt.detach(), "SmoothnessLoss")
        ws, scale = _ws if _ws is not None else self._workspace(gt, c, h, w)
        m = torch.empty((h, w), device=gt.device, dtype=torch.float32)
        with torch.cuda.device(gt.device):
            _native._check(_lib.tsl_smoothness_mask(gt.data_ptr(), c, h, w, scale, float(self.quantile), ws.data_ptr(), ws.numel(), m.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "SmoothnessLoss mask")
        return m

    def forward(self, img: torch.Tensor, img_gt: torch.Tensor) -> torch.Tensor:
        c, h, w, x, gt = _aux_prepare(img, img_gt, "SmoothnessLoss")
        ws = self._workspace(gt, c, h, w)
        return _ScharrSmoothness.apply(x, self.mask(gt, ws), c, h, w, ws[0])


dogLoss = DoGLoss()                  # the module-level instances of trainer_utils.py:350-351
smoothnessLoss = SmoothnessLoss()


# ---- the down-sampler of render_up_scale (include/ts_loss.h: tsl_downsample_*; csrc/resample.hip) ----------------------------------------
_lib.tsl_downsample_forward.restype = C.c_int
_lib.tsl_downsample_forward.argtypes = [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _fp]
_lib.tsl_downsample_backward.restype = C.c_int
_lib.tsl_downsample_backward.argtypes = [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _fp]
for _name in ("tsl_downsample_forward_planes", "tsl_downsample_backward_planes"):
    getattr(_lib, _name).restype = C.c_int
    getattr(_lib, _name).argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), _fp]


class _Downsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, w):
        xc = x.contiguous()
        lead, (H, W) = xc.shape[:-2], xc.shape[-2:]
        c = 1
        for d in lead:
            c *= int(d)
        out = torch.empty(tuple(lead) + (h, w), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _native._check(_lib.tsl_downsample_forward(xc.data_ptr(), c, H, W, h, w, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "downsample_bilinear")
        ctx.dims = (c, H, W, h, w, tuple(xc.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        c, H, W, h, w, shape = ctx.dims
        gc = g.contiguous()
        gin = torch.empty(shape, device=g.device, dtype=torch.float32)
        with torch.cuda.device(g.device):
            _native._check(_lib.tsl_downsample_backward(gc.data_ptr(), c, H, W, h, w, gin.data_ptr(), torch.cuda.current_stream().cuda_stream), "downsample_bilinear backward")
        return gin, None, None


def _plane_ptrs(t: torch.Tensor, hw: int):
    lead = t.numel() // hw
    return [t.data_ptr() + 4 * hw * k for k in range(lead)]


class _DownsampleMany(torch.autograd.Function):
    """downsample_bilinear of several tensors with the same (H, W) in ONE launch each way (include/ts_loss.h: tsl_downsample_*_planes)."""

    @staticmethod
    def forward(ctx, h, w, *xs):
        xs = [x.contiguous() for x in xs]
        H, W = xs[0].shape[-2:]
        outs = [torch.empty(tuple(x.shape[:-2]) + (h, w), device=x.device, dtype=torch.float32) for x in xs]
        src = sum((_plane_ptrs(x, H * W) for x in xs), [])
        dst = sum((_plane_ptrs(o, h * w) for o in outs), [])
        n = len(src)
        with torch.cuda.device(xs[0].device):
            _native._check(_lib.tsl_downsample_forward_planes(n, (C.c_void_p * n)(*src), H, W, h, w, (C.c_void_p * n)(*dst),
                                                              torch.cuda.current_stream().cuda_stream), "downsample_bilinear")
        ctx.dims = (H, W, h, w, [tuple(x.shape) for x in xs])
        ctx.set_materialize_grads(False)  # an output nobody differentiates (depth / normal without the geometry loss) hands its input no gradient
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        H, W, h, w, shapes = ctx.dims
        live = [(i, g.contiguous()) for i, g in enumerate(gs) if g is not None]
        gins = [None] * len(gs)
        if live:
            dev = live[0][1].device
            for i, _ in live:
                gins[i] = torch.empty(shapes[i], device=dev, dtype=torch.float32)
            src = sum((_plane_ptrs(g, h * w) for _, g in live), [])
            dst = sum((_plane_ptrs(gins[i], H * W) for i, _ in live), [])
            n = len(src)
            with torch.cuda.device(dev):
                _native._check(_lib.tsl_downsample_backward_planes(n, (C.c_void_p * n)(*src), H, W, h, w, (C.c_void_p * n)(*dst),
                                                                   torch.cuda.current_stream().cuda_stream), "downsample_bilinear backward")
        return (None, None) + tuple(gins)


def downsample_bilinear_many(xs, size):
    """[downsample_bilinear(x, size) for x in xs] for tensors of one (H, W) -- the render, depth and normal images of a step -- in one launch each
    way instead of one per tensor (at 800 x 800 a launch is ~9 us of latency for ~3 of work).  An output without an incoming gradient gives its
    input none (not zeros): the rasterizer's backward then takes its colour-only form."""
    h, w = int(size[0]), int(size[1])
    xs = list(xs)
    if not xs:
        return []
    for x in xs:
        if not x.is_cuda or x.dtype != torch.float32:
            raise RuntimeError("downsample_bilinear (MI355X build) needs float32 tensors on a HIP device; there is no CPU fallback")
        if tuple(x.shape[-2:]) != tuple(xs[0].shape[-2:]):
            raise ValueError("downsample_bilinear_many: the tensors must share their last two dimensions")
    return list(_DownsampleMany.apply(h, w, *xs))


def downsample_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """`F.interpolate(x[None], size=size, mode="bilinear")[0]` for an INTEGER down-sampling factor in both directions -- the resize that follows a
    render at render_up_scale x the camera's resolution (VanillaTS_model.py:649-656) -- as one HBM-bound HIP kernel each way; the backward gathers
    (no atomics: run-to-run identical, unlike torch's upsample_bilinear2d_backward).  x: (..., H, W) float32 on the HIP device."""
    h, w = int(size[0]), int(size[1])
    if not x.is_cuda or x.dtype != torch.float32:
        raise RuntimeError("downsample_bilinear (MI355X build) needs a float32 tensor on a HIP device; there is no CPU fallback")
    return _Downsample.apply(x, h, w)
