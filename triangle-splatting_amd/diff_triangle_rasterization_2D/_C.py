"""`_C` of the drop-in `diff_triangle_rasterization_2D` package: the two entry points the reference exports
through pybind (R2D/ext.cpp:6-8), here bound with ctypes onto the C ABI of libts2d.so (include/ts2d.h).

    rasterize_triangles(image_width, image_height, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos,
                        sh_degree, gamma, scale_modifier, background_depth, background, vertex, shs, feature,
                        opacity, back_culling, rich_info, debug)
        -> (num_rendered, out_feature, radii, depth, normal, contrib_sum, contrib_max,
            geometryBuffer, binningBuffer, imageBuffer)          # R2D/src/extension_interface.cu:19-152
    rasterize_triangles_backward(tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma,
                        scale_modifier, background_depth, background, vertex, shs, feature, opacity,
                        num_rendered, radii, geometryBuffer, binningBuffer, imageBuffer, dL_dout_feature,
                        dL_dout_depth, dL_dout_normal, rich_info, debug)
        -> (dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity)   # extension_interface.cu:154-260

Same positional signatures, same argument checks and RuntimeErrors, same ownership (the callee allocates every
output on vertex.device; the three uint8 buffers are opaque).  torch is used only for device memory (caching
allocator) and the current stream.  There is NO CPU or eager fallback: if libts2d.so is missing or the tensors
are not on a HIP device, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libts2d.so")
# Measurement / triage only (tools/, tests/triage/, the lab-library tests): a statistics or lab build of the same C ABI
# (tools/bin/libts2d_stats.so, tools/bin/libts2d_lab.so) can be loaded in place of the product library.  The product library
# itself reads no environment variable.
_LIB_PATH = os.environ.get("TS2D_LIBRARY_PATH") or _LIB_PATH

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} not found: build it with `python triangle-splatting_amd/build.py` (hipcc, gfx950). "
        "The MI355X rasterizer has no CPU fallback."
    )
_lib = C.CDLL(_LIB_PATH)

# ---- the compiled binding (bindings/ts2d_torch_ext.cpp -> bindings/_ts2d_torch_C.so) is the default since round 6 ---------------------------------
# The two hot entry points below cost 0.3-0.4 ms of host time per forward + backward through ctypes (argument marshalling, ~15 torch allocations from
# Python) -- what bounds every scene below ~100 k triangles.  The torch extension built by __graft_entry__.build() does the same
# work in C++ on the same libts2d.so; it is used whenever it exists.  ctypes remains (a) the fallback when the extension was not built, (b) the path
# of every measurement that swaps the library (TS2D_LIBRARY_PATH: the extension is linked against the product library), (c) selectable with
# TS2D_BINDING=ctypes (tests/test_binding_gpu.py runs the package through both).  Everything that is not on the per-step path (profile hooks,
# capacity hints, sh_grad_expand, forward_status) stays on ctypes: same library instance, loaded once.
_ext = None
_EXT_PATH = os.path.join(os.path.dirname(_HERE), "bindings", "_ts2d_torch_C.so")
if os.environ.get("TS2D_BINDING", "") != "ctypes" and not os.environ.get("TS2D_LIBRARY_PATH") and os.path.exists(_EXT_PATH):
    try:
        import importlib.util as _ilu
        _spec = _ilu.spec_from_file_location("_ts2d_torch_C", _EXT_PATH)
        _mod = _ilu.module_from_spec(_spec)
        _spec.loader.exec_module(_mod)
        if hasattr(_mod, "rasterize_triangles_ex") and hasattr(_mod, "rasterize_triangles_backward_ex"):
            _ext = _mod
    except (ImportError, OSError) as _e:  # a stale build against another torch: fall back, loudly
        import warnings
        warnings.warn(f"{_EXT_PATH} could not be loaded ({_e}); using the ctypes binding (rebuild with bindings/build_torch_ext.py --force)")


def binding() -> str:
    """'compiled' or 'ctypes': which binding rasterize_triangles / rasterize_triangles_backward go through."""
    return "compiled" if _ext is not None else "ctypes"


FLAG_BACK_CULLING, FLAG_RICH_INFO, FLAG_DEBUG, FLAG_USE_SHS, FLAG_3D, FLAG_SH_FACTORED = 1, 2, 4, 8, 16, 32
MAX_CHANNELS = 3

_fp = C.c_void_p


class _Camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("viewmatrix", _fp), ("projmatrix", _fp), ("campos", _fp)]


class _Geometry(C.Structure):
    _fields_ = [("P", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32), ("C", C.c_int32),
                ("gamma", C.c_float), ("scale_modifier", C.c_float), ("background_depth", C.c_float),
                ("background", _fp), ("vertex", _fp), ("shs", _fp), ("feature", _fp), ("opacity", _fp), ("background_depth_dev", _fp)]


class _ForwardOut(C.Structure):
    _fields_ = [("out_feature", _fp), ("depth", _fp), ("normal", _fp), ("contrib_sum", _fp), ("contrib_max", _fp)]


class _LossGrads(C.Structure):
    _fields_ = [("dL_dout_feature", _fp), ("dL_dout_depth", _fp), ("dL_dout_normal", _fp)]


class _BackwardOut(C.Structure):
    _fields_ = [("dL_dvertex", _fp), ("dL_dcenter2D", _fp), ("dL_dshs", _fp), ("dL_dfeature", _fp),
                ("dL_dopacity", _fp)]


class _State(C.Structure):
    _fields_ = [("geometry", _fp), ("geometry_bytes", C.c_size_t), ("binning", _fp), ("binning_bytes", C.c_size_t),
                ("image", _fp), ("image_bytes", C.c_size_t)]


_lib.ts2d_version.restype = C.c_char_p
_lib.ts2d_last_error.restype = C.c_char_p
_lib.ts2d_geometry_state_bytes.restype = C.c_size_t
_lib.ts2d_geometry_state_bytes.argtypes = [C.c_int32]
_lib.ts2d_binning_state_bytes.restype = C.c_size_t
_lib.ts2d_binning_state_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
_lib.ts2d_image_state_bytes.restype = C.c_size_t
_lib.ts2d_image_state_bytes.argtypes = [C.c_int32, C.c_int32]
_lib.ts2d_backward_scratch_bytes.restype = C.c_size_t
_lib.ts2d_backward_scratch_bytes.argtypes = [C.c_int32]
_lib.ts2d_forward_bin.restype = C.c_int
_lib.ts2d_forward_bin.argtypes = [C.POINTER(_Camera), C.POINTER(_Geometry), C.c_uint32, _fp, C.POINTER(_State),
                                  C.POINTER(C.c_int64), _fp]
_lib.ts2d_forward_render.restype = C.c_int
_lib.ts2d_forward_render.argtypes = [C.POINTER(_Camera), C.POINTER(_Geometry), C.c_uint32, C.c_int64,
                                     C.POINTER(_State), C.POINTER(_ForwardOut), _fp]
_lib.ts2d_forward.restype = C.c_int
_lib.ts2d_forward.argtypes = [C.POINTER(_Camera), C.POINTER(_Geometry), C.c_uint32, _fp, C.POINTER(_State), C.c_int64,
                              C.POINTER(_ForwardOut), _fp]
_lib.ts2d_forward_status.restype = C.c_int
_lib.ts2d_forward_status.argtypes = [C.POINTER(_State), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64), _fp]
_lib.ts2d_backward.restype = C.c_int
_lib.ts2d_backward.argtypes = [C.POINTER(_Camera), C.POINTER(_Geometry), C.c_uint32, C.c_int64, _fp,
                               C.POINTER(_State), C.POINTER(_LossGrads), _fp, C.c_size_t, C.POINTER(_BackwardOut), _fp]
_lib.ts2d_backward_ranged.restype = C.c_int
_lib.ts2d_backward_ranged.argtypes = [C.POINTER(_Camera), C.POINTER(_Geometry), C.c_uint32, C.c_int64, _fp,
                                      C.POINTER(_State), C.POINTER(_LossGrads), _fp, C.c_size_t, C.POINTER(_BackwardOut), C.c_int32, C.POINTER(_fp), _fp]
_lib.ts2d_backward_range_rows.restype = C.c_int32
_lib.ts2d_backward_range_rows.argtypes = [C.c_int32, C.c_int32]
_lib.ts2d_sh_grad_expand.restype = C.c_int
_lib.ts2d_sh_grad_expand.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp]
_lib.ts2d_binning_capacity.restype = C.c_int64
_lib.ts2d_binning_capacity.argtypes = [C.c_size_t, C.c_int32, C.c_int32]
_lib.ts2d_instance_capacity_hint.restype = C.c_int64
_lib.ts2d_instance_capacity_hint.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint32]
_lib.ts2d_set_capacity_hint_key.restype = None
_lib.ts2d_set_capacity_hint_key.argtypes = [C.c_uint64]
_lib.ts2d_speculative_overflow_count.restype = C.c_uint64
_lib.ts2d_speculative_overflow_count.argtypes = []
_lib.ts2d_forward_speculative.restype = C.c_int
_lib.ts2d_forward_speculative.argtypes = [C.POINTER(_Camera), C.POINTER(_Geometry), C.c_uint32, _fp, C.POINTER(_State), C.POINTER(_ForwardOut),
                                          C.POINTER(C.c_int64), _fp]
_lib.ts2d_profile_enable.argtypes = [C.c_int]
_lib.ts2d_profile_only.argtypes = [C.c_char_p]
_lib.ts2d_profile_read.restype = C.c_int
_lib.ts2d_profile_read.argtypes = [C.c_int32, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int64)]


def set_capacity_hint_key(key: int) -> None:
    """Names the stream of views the calling thread's next forwards belong to (ts2d_set_capacity_hint_key): the speculative forward sizes its
    binning buffer from the history of (device, variant, image size, key).  Train and evaluation cameras of one size, or two models in one
    process, should use different keys; the default is 0."""
    _lib.ts2d_set_capacity_hint_key(int(key) & 0xFFFFFFFFFFFFFFFF)


def speculative_overflows() -> int:
    """How many speculative forwards of this process guessed too small a binning buffer and rendered a second time (ts2d.h)."""
    return int(_lib.ts2d_speculative_overflow_count())


def version() -> str:
    return _lib.ts2d_version().decode()


def library_path() -> str:
    return _LIB_PATH


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what}: {_lib.ts2d_last_error().decode()} (ts2d error {rc})")


def _ptr(t):
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _use_shs(shs: torch.Tensor, feature: torch.Tensor) -> bool:
    # R2D/src/extension_interface.cu:44
    return feature.dim() <= 1 or (feature.size(0) == 0 and shs.size(0) > 0)


def _require_device(vertex: torch.Tensor):
    if not vertex.is_cuda:
        raise RuntimeError(
            "diff_triangle_rasterization_2D (MI355X build) needs tensors on a HIP device; there is no CPU fallback"
        )


def _contiguous_or_raise(*tensors):
    # extension_interface.cu:77-81, 193-199
    for t in tensors:
        if t is not None and not t.is_contiguous():
            raise RuntimeError("input tensors must be contiguous")


def _f32_or_raise(*tensors):
    for t in tensors:
        if t is not None and t.numel() > 0 and t.dtype != torch.float32:
            raise RuntimeError("expected scalar type Float")  # what data_ptr<float>() raises in the reference


def _marshal(W, H, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma, scale_modifier,
             background_depth, background, vertex, shs, feature, opacity, use_shs, Cn, M):
    cam = _Camera(int(W), int(H), float(tan_fovx), float(tan_fovy), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos))
    # background_depth: a float like the reference's binding takes -- or a one-element float32 tensor on the device (what the reference's
    # model computes every step, VanillaTS_model.py:623), handed to the kernels as a pointer: no device synchronisation for the conversion
    bg_dev = None
    if isinstance(background_depth, torch.Tensor):
        if not (background_depth.is_cuda and background_depth.dtype == torch.float32 and background_depth.numel() == 1):
            raise RuntimeError("background_depth must be a float or a one-element float32 tensor on the HIP device")
        bg_dev, background_depth = background_depth.data_ptr(), 0.0
    geom = _Geometry(int(vertex.size(0)), int(sh_degree), int(M), int(Cn), float(gamma), float(scale_modifier),
                     float(background_depth), _ptr(background), _ptr(vertex), _ptr(shs) if use_shs else None,
                     None if use_shs else _ptr(feature), _ptr(opacity), bg_dev)
    return cam, geom


def _state(geometryBuffer, binningBuffer, imageBuffer) -> _State:
    return _State(_ptr(geometryBuffer), geometryBuffer.numel(), _ptr(binningBuffer), binningBuffer.numel(),
                  _ptr(imageBuffer), imageBuffer.numel())


def rasterize_triangles(image_width, image_height, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma,
                        scale_modifier, background_depth, background, vertex, shs, feature, opacity, back_culling,
                        rich_info, debug, *, variant=2, instance_capacity=None):
    """`variant=3` selects the 3D rasterizer (TS2D_FLAG_3D; used by the sibling package diff_triangle_rasterization_3D).
    `instance_capacity` (an int > 0) selects the SYNC-FREE forward (ts2d_forward): the binning state is sized for that many tile
    instances, nothing is read back, and the returned `num_rendered` is the capacity (it only sizes the state for the backward
    call); whether the true count fitted is reported by `forward_status`.  Default None = the reference's sequence with its one
    blocking read of num_rendered."""
    if _ext is not None:
        bg_t = background_depth if isinstance(background_depth, torch.Tensor) else None
        return _ext.rasterize_triangles_ex(int(image_width), int(image_height), tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, int(sh_degree), gamma,
                                           scale_modifier, 0.0 if bg_t is not None else float(background_depth), background, vertex, shs, feature, opacity,
                                           bool(back_culling), bool(rich_info), bool(debug), int(variant),
                                           int(instance_capacity) if (instance_capacity is not None and vertex.size(0) > 0) else 0, bg_t)
    P = vertex.size(0)
    H, W = int(image_height), int(image_width)
    use_shs = _use_shs(shs, feature)
    Cn = 3 if use_shs else feature.size(1)
    M = shs.size(1) if (shs.size(0) != 0 and shs.dim() >= 2) else 0

    # R2D/src/extension_interface.cu:53-81
    if vertex.dim() != 3 or vertex.size(1) != 3 or vertex.size(2) != 3:
        raise RuntimeError("vertex must have dimensions (num_points, 3, 3)")
    if not use_shs and feature.dim() != 2:
        raise RuntimeError("feature must have dimensions (num_points, num_channels)")
    if use_shs and shs.dim() != 3:
        raise RuntimeError("shs must have dimensions (num_points, (1 + sh_degree) ** 2, 3)")
    if Cn > MAX_CHANNELS:
        raise RuntimeError("feature's num_channels can't be larger than MAX_CHANNELS")
    if Cn != background.size(0):
        raise RuntimeError("background must have the same number of channels as feature")
    if gamma < 0.0:
        raise RuntimeError("gamma must be larger than 0")
    if variant == 3:  # R3D/src/extension_interface.cu:82-92 takes .contiguous() of every input instead of raising
        viewmatrix, projmatrix, campos, background, vertex, shs, feature, opacity = (
            t.contiguous() for t in (viewmatrix, projmatrix, campos, background, vertex, shs, feature, opacity))
    else:
        _contiguous_or_raise(viewmatrix, projmatrix, campos, background, vertex, shs, feature, opacity)
    _require_device(vertex)
    _f32_or_raise(viewmatrix, projmatrix, campos, background, vertex, shs if use_shs else None,
                  None if use_shs else feature, opacity)

    dev = vertex.device
    f32 = dict(device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):  # OptionalCUDAGuard, extension_interface.cu:83
        stream = torch.cuda.current_stream().cuda_stream
        # every element of these is written by the library (ts2d.h), so no zero-fill pass is needed when P > 0
        alloc = torch.zeros if P == 0 else torch.empty
        out_feature = alloc((Cn, H, W), **f32)
        radii = alloc((P,), device=dev, dtype=torch.int32)
        if rich_info:
            depth = alloc((H, W), **f32)
            normal = alloc((3, H, W), **f32)
            contrib_sum = alloc((P,), **f32)
            contrib_max = alloc((P,), **f32)
        else:
            depth = torch.empty((0,), **f32)
            normal = torch.empty((0,), **f32)
            contrib_sum = torch.empty((0,), **f32)
            contrib_max = torch.empty((0,), **f32)
        u8 = dict(device=dev, dtype=torch.uint8)
        if P == 0:  # extension_interface.cu:130: zero images, empty state
            return (0, out_feature, radii, depth, normal, contrib_sum, contrib_max, torch.empty((0,), **u8),
                    torch.empty((0,), **u8), torch.empty((0,), **u8))

        flags = ((FLAG_BACK_CULLING if back_culling else 0) | (FLAG_RICH_INFO if rich_info else 0) |
                 (FLAG_DEBUG if debug else 0) | (FLAG_USE_SHS if use_shs else 0) | (FLAG_3D if variant == 3 else 0))
        cam, geom = _marshal(W, H, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma, scale_modifier,
                             background_depth, background, vertex, shs, feature, opacity, use_shs, Cn, M)
        geometryBuffer = torch.empty((_lib.ts2d_geometry_state_bytes(P),), **u8)
        imageBuffer = torch.empty((_lib.ts2d_image_state_bytes(W, H),), **u8)
        if instance_capacity is not None:
            cap = int(instance_capacity)
            binningBuffer = torch.empty((_lib.ts2d_binning_state_bytes(cap, W, H),), **u8)
            st = _state(geometryBuffer, binningBuffer, imageBuffer)
            out = _ForwardOut(_ptr(out_feature), _ptr(depth), _ptr(normal), _ptr(contrib_sum), _ptr(contrib_max))
            _check(_lib.ts2d_forward(C.byref(cam), C.byref(geom), flags, _ptr(radii), C.byref(st), cap, C.byref(out), stream),
                   "rasterize_triangles")
            return (cap, out_feature, radii, depth, normal, contrib_sum, contrib_max, geometryBuffer, binningBuffer, imageBuffer)
        # The reference's sequence (num_rendered comes back to the host, rasterizer.cu:189-191) without its stall: the binning buffer is
        # sized from what recent forwards of this image size rendered (x 1.25), EVERYTHING is queued for that capacity, and only then does
        # the host wait for the exact count, which the GPU publishes ~0.1 ms into the forward (ts2d_forward_speculative).  Without a
        # history (first call) or when the guess was too small (nothing was emitted then), the second half runs again with the exact size.
        cap_guess = int(_lib.ts2d_instance_capacity_hint(P, W, H, flags))
        binningBuffer = torch.empty((_lib.ts2d_binning_state_bytes(cap_guess, W, H) if cap_guess > 0 else 0,), **u8)
        st = _state(geometryBuffer, binningBuffer, imageBuffer)
        out = _ForwardOut(_ptr(out_feature), _ptr(depth), _ptr(normal), _ptr(contrib_sum), _ptr(contrib_max))
        n = C.c_int64(0)
        _check(_lib.ts2d_forward_speculative(C.byref(cam), C.byref(geom), flags, _ptr(radii), C.byref(st), C.byref(out), C.byref(n), stream),
               "rasterize_triangles")
        num_rendered = int(n.value)
        if cap_guess <= 0 or num_rendered > _lib.ts2d_binning_capacity(binningBuffer.numel(), W, H):
            binningBuffer = torch.empty((_lib.ts2d_binning_state_bytes(num_rendered, W, H),), **u8)
            st = _state(geometryBuffer, binningBuffer, imageBuffer)
            _check(_lib.ts2d_forward_render(C.byref(cam), C.byref(geom), flags, num_rendered, C.byref(st), C.byref(out), stream),
                   "rasterize_triangles")
    return (num_rendered, out_feature, radii, depth, normal, contrib_sum, contrib_max, geometryBuffer, binningBuffer,
            imageBuffer)


def rasterize_triangles_backward(tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma, scale_modifier,
                                 background_depth, background, vertex, shs, feature, opacity, num_rendered, radii,
                                 geometryBuffer, binningBuffer, imageBuffer, dL_dout_feature, dL_dout_depth,
                                 dL_dout_normal, rich_info, debug, *, variant=2, sh_factored=False, out=None, range_events=None):
    """`sh_factored=True` (SH mode only; TS2D_FLAG_SH_FACTORED): dL_dshs is not formed (returned as None) and the fourth
    result holds the clamp-masked colour gradient dL_dRGB (P, 3) for `sh_grad_expand` -- see parallel.py.
    `out`: optional dict of preallocated contiguous float32 device tensors ("vertex" (P,3,3), "center2D" (P,2), "opacity" (P,1),
    "color" = dL_dshs (P,M,3) or dL_dfeature (P,C)) that the library writes instead of fresh allocations (parallel.GradBucket).
    `range_events`: a list of K torch.cuda.Event (each recorded at least once before): the per-triangle kernel runs as K launches over
    consecutive triangle ranges of `backward_range_rows(P, K)` rows and event k is recorded behind range k (ts2d_backward_ranged)."""
    if _ext is not None:
        bg_t = background_depth if isinstance(background_depth, torch.Tensor) else None
        o = out or {}
        return _ext.rasterize_triangles_backward_ex(tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, int(sh_degree), gamma, scale_modifier,
                                                    0.0 if bg_t is not None else float(background_depth), background, vertex, shs, feature, opacity,
                                                    int(num_rendered), radii, geometryBuffer, binningBuffer, imageBuffer, dL_dout_feature, dL_dout_depth,
                                                    dL_dout_normal, bool(rich_info), bool(debug), int(variant), bool(sh_factored), o.get("vertex"),
                                                    o.get("center2D"), o.get("color"), o.get("opacity"), bg_t,
                                                    [int(e.cuda_event) for e in range_events] if range_events else [])
    P = vertex.size(0)
    H, W = dL_dout_feature.size(1), dL_dout_feature.size(2)  # extension_interface.cu:182-183
    use_shs = _use_shs(shs, feature)
    Cn = 3 if use_shs else feature.size(1)
    M = shs.size(1) if (shs.size(0) != 0 and shs.dim() >= 2) else 0
    if variant == 3:  # R3D/src/extension_interface.cu:186-206: .contiguous() instead of the 2D module's error
        (viewmatrix, projmatrix, campos, background, vertex, shs, feature, opacity, radii, dL_dout_feature, dL_dout_depth,
         dL_dout_normal) = (t.contiguous() for t in (viewmatrix, projmatrix, campos, background, vertex, shs, feature, opacity,
                                                     radii, dL_dout_feature, dL_dout_depth, dL_dout_normal))
    _contiguous_or_raise(viewmatrix, projmatrix, campos, background, vertex, shs, feature, opacity, radii, geometryBuffer,
                         binningBuffer, imageBuffer, dL_dout_feature, dL_dout_depth, dL_dout_normal)
    _require_device(vertex)
    _f32_or_raise(dL_dout_feature, dL_dout_depth, dL_dout_normal)

    dev = vertex.device
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        alloc = torch.zeros if P == 0 else torch.empty  # every element is written by the library when P > 0
        opts = dict(device=dev, dtype=vertex.dtype)
        out = out or {}

        def placed(name, shape, fallback):
            t = out.get(name)
            if t is None:
                return fallback(shape, **opts)
            if tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.dtype != torch.float32 or t.device != dev:
                raise RuntimeError(f"preallocated gradient output '{name}' must be a contiguous float32 {tuple(shape)} tensor on {dev}")
            if P == 0:
                t.zero_()
            return t

        dL_dvertex = placed("vertex", (P, 3, 3), alloc)
        dL_dcenter2D = placed("center2D", (P, 2), alloc)
        sh_factored = bool(sh_factored and use_shs)
        if sh_factored:
            dL_dshs = None
        else:
            dL_dshs = placed("color", (P, M, 3), alloc) if use_shs else torch.zeros((P, M, 3), **opts)
        dL_dfeature = alloc((P, Cn), **opts) if use_shs else placed("color", (P, Cn), alloc)
        dL_dopacity = placed("opacity", (P, 1), alloc)
        if P == 0:
            return dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity
        flags = ((FLAG_RICH_INFO if rich_info else 0) | (FLAG_DEBUG if debug else 0) | (FLAG_USE_SHS if use_shs else 0) |
                 (FLAG_3D if variant == 3 else 0) | (FLAG_SH_FACTORED if sh_factored else 0))
        cam, geom = _marshal(W, H, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma, scale_modifier,
                             background_depth, background, vertex, shs, feature, opacity, use_shs, Cn, M)
        st = _state(geometryBuffer, binningBuffer, imageBuffer)
        loss = _LossGrads(_ptr(dL_dout_feature), _ptr(dL_dout_depth) if rich_info else None,
                          _ptr(dL_dout_normal) if rich_info else None)
        scratch = torch.empty((_lib.ts2d_backward_scratch_bytes(P),), device=dev, dtype=torch.uint8)
        scratch_bytes = scratch.numel()
        out = _BackwardOut(_ptr(dL_dvertex), _ptr(dL_dcenter2D), _ptr(dL_dshs), _ptr(dL_dfeature), _ptr(dL_dopacity))
        if range_events:
            handles = (_fp * len(range_events))(*[int(e.cuda_event) for e in range_events])
            _check(_lib.ts2d_backward_ranged(C.byref(cam), C.byref(geom), flags, int(num_rendered), _ptr(radii), C.byref(st), C.byref(loss), _ptr(scratch),
                                             scratch_bytes, C.byref(out), len(range_events), handles, stream), "rasterize_triangles_backward")
        else:
            _check(_lib.ts2d_backward(C.byref(cam), C.byref(geom), flags, int(num_rendered), _ptr(radii), C.byref(st),
                                      C.byref(loss), _ptr(scratch), scratch_bytes, C.byref(out), stream),
                   "rasterize_triangles_backward")
    return dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity


def backward_range_rows(P: int, num_ranges: int) -> int:
    """Rows per triangle range of a ranged backward (ts2d_backward_range_rows): ceil(P / K) rounded up to a multiple of 64."""
    return int(_lib.ts2d_backward_range_rows(int(P), int(num_ranges)))


def forward_status(P, W, H, geometryBuffer, imageBuffer):
    """(overflowed, num_rendered) of the last sync-free forward on these state buffers (ts2d_forward_status; one blocking read)."""
    st = _State(_ptr(geometryBuffer), geometryBuffer.numel(), None, 0, _ptr(imageBuffer), imageBuffer.numel())
    over, n = C.c_int32(0), C.c_int64(0)
    with torch.cuda.device(imageBuffer.device):
        _check(_lib.ts2d_forward_status(C.byref(st), int(P), int(W), int(H), C.byref(over), C.byref(n), torch.cuda.current_stream().cuda_stream),
               "forward_status")
    return bool(over.value), int(n.value)


def sh_grad_expand(vertex, campos, dL_dcolor, sh_degree, M, out=None):
    """dL_dshs (P, M, 3) = sum over views v of basis(normalize(centroid - campos[v])) x dL_dcolor[v]  (ts2d_sh_grad_expand).
    vertex (P,3,3), campos (V,3), dL_dcolor (V,P,3): contiguous float32 on one HIP device."""
    _require_device(vertex)
    _contiguous_or_raise(vertex, campos, dL_dcolor, out)
    _f32_or_raise(vertex, campos, dL_dcolor, out)
    P, V = vertex.size(0), campos.size(0)
    if dL_dcolor.shape != (V, P, 3):
        raise RuntimeError("dL_dcolor must have dimensions (num_views, num_points, 3)")
    with torch.cuda.device(vertex.device):
        if out is None:
            out = (torch.zeros if P == 0 else torch.empty)((P, int(M), 3), device=vertex.device, dtype=torch.float32)
        _check(_lib.ts2d_sh_grad_expand(P, int(sh_degree), int(M), V, _ptr(vertex), _ptr(campos), _ptr(dL_dcolor), _ptr(out),
                                        torch.cuda.current_stream().cuda_stream), "sh_grad_expand")
    return out


# ---- timing hooks used by bench.py (not part of the reference's surface) ----------------------------------------
def profile_enable(on: bool):
    _lib.ts2d_profile_enable(1 if on else 0)


def profile_only(name: str = ""):
    _lib.ts2d_profile_only(name.encode() if name else None)


def profile_reset():
    _lib.ts2d_profile_reset()


def profile_read():
    rows, i = [], 0
    name = C.create_string_buffer(64)
    ms, n = C.c_double(0), C.c_int64(0)
    while _lib.ts2d_profile_read(i, name, 64, C.byref(ms), C.byref(n)) == 0:
        rows.append((name.value.decode(), ms.value, n.value))
        i += 1
    return rows
