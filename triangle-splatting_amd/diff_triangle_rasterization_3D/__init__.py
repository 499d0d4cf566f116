"""MI355X-native drop-in for the reference's `diff_triangle_rasterization_3D` Python package (rasterizer_type "3D").

Public surface as in R3D/diff_triangle_rasterization_3D/__init__.py (R3D = submodules/diff-triangle-rasterization-3D):
`TriangleRasterizationSettings` (:28-46) and `TriangleRasterizer` (:167-187), selected by
src/diff_recon/renderer/triangle_renderer.py:5-6 when `rasterizer_type == "3D"`.  The reference's two Python modules
differ only in where `.contiguous()` is applied (the 3D extension makes its inputs contiguous itself instead of
raising, R3D/src/extension_interface.cu:82-92), so this package reuses the 2D package's classes and flips the native
variant: same libts2d.so, entry points called with TS2D_FLAG_3D (include/ts2d.h), which selects preprocess3d.hip and
render3d.hip (ray/plane intersection, 3D barycentrics, unnormalised normals).
"""
from __future__ import annotations

from diff_triangle_rasterization_2D import (TriangleRasterizationSettings, TriangleRasterizer as _Rasterizer2D,
                                            _RasterizeTriangles as _Rasterize2D, _C)


class _RasterizeTriangles(_Rasterize2D):
    _variant = 3


class TriangleRasterizer(_Rasterizer2D):
    _function = _RasterizeTriangles


__all__ = ["TriangleRasterizationSettings", "TriangleRasterizer"]
