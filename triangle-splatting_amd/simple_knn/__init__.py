"""MI355X-native drop-in for the reference's `simple_knn` package (submodules/simple-knn/simple_knn/__init__.py:6-23):

    distCUDA2(points)                     mean squared distance of each point to its 3 nearest neighbours
    nearestNeighbor(points, batch_size)   index of the nearest point outside the point's own batch group (uint32)

Callers in the reference: src/diff_recon/models/model_utils.py:36, src/diff_recon/utils/vis_utils.py:48,
src/diff_recon/trainers/trainer_utils.py:339-340.  Native code: libts2d.so (include/ts_knn.h, csrc/knn.hip); there is no
CPU fallback.  Argument checks and messages follow submodules/simple-knn/interface.cu:8-11,30-37.
"""
from __future__ import annotations

import ctypes as C

import torch

from diff_triangle_rasterization_2D import _C as _native

_lib = _native._lib
_fp = C.c_void_p
_lib.tsk_workspace_bytes.restype = C.c_size_t
_lib.tsk_workspace_bytes.argtypes = [C.c_int32]
_lib.tsk_mean_dist3.restype = C.c_int
_lib.tsk_mean_dist3.argtypes = [C.c_int32, _fp, _fp, _fp, C.c_size_t, _fp]
_lib.tsk_nearest_other.restype = C.c_int
_lib.tsk_nearest_other.argtypes = [C.c_int32, C.c_int32, _fp, _fp, _fp, C.c_size_t, _fp]


def _prepare(points: torch.Tensor):
    if not points.is_cuda:
        raise RuntimeError("simple_knn (MI355X build) needs tensors on a HIP device; there is no CPU fallback")
    if points.dtype != torch.float32:
        raise RuntimeError("expected scalar type Float")
    return points.contiguous()  # interface.cu:21,46


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.dim() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    pts = _prepare(points)
    P = pts.size(0)
    with torch.cuda.device(pts.device):
        means = torch.zeros((P,), device=pts.device, dtype=torch.float32)
        if P == 0:
            return means
        nbytes = _lib.tsk_workspace_bytes(P)
        ws = torch.empty((nbytes,), device=pts.device, dtype=torch.uint8)
        _native._check(_lib.tsk_mean_dist3(P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(), nbytes,
                                           torch.cuda.current_stream().cuda_stream), "distCUDA2")
    return means


def nearestNeighbor(points: torch.Tensor, batch_size: int = 1) -> torch.Tensor:
    if batch_size <= 0:
        raise RuntimeError("batch_size must be greater than 0")
    if points.dim() != 2 or points.size(1) != 3 or points.size(0) % batch_size != 0:
        raise RuntimeError("points must have dimensions (num_points, 3) and num_points % batch_size == 0, "
                           f"where batch_size = {batch_size}")
    pts = _prepare(points)
    P = pts.size(0)
    with torch.cuda.device(pts.device):
        out = torch.zeros((P,), device=pts.device, dtype=torch.int32)
        if P > 0:
            nbytes = _lib.tsk_workspace_bytes(P)
            ws = torch.empty((nbytes,), device=pts.device, dtype=torch.uint8)
            _native._check(_lib.tsk_nearest_other(P, int(batch_size), pts.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes,
                                                  torch.cuda.current_stream().cuda_stream), "nearestNeighbor")
    return out.view(torch.uint32)  # the reference returns kUInt32 (interface.cu:44)


__all__ = ["distCUDA2", "nearestNeighbor"]
