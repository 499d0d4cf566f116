"""ctypes/numpy front-end of the CPU oracle (oracle/ts2d_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package never imports this module.

The functions mirror the reference's pybind entry points `rasterize_triangles` /
`rasterize_triangles_backward` (R2D/ext.cpp:6-8, R2D/src/extension_interface.h:7-62) with numpy
arrays in place of torch tensors.  Parity status of the oracle itself: see the header of
ts2d_oracle.c (pinned against the reference's own kernels built for gfx950, oracle/build_ref.py + tests/test_reference_gpu.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libts2d_oracle.so")
_lib = None

_FIELDS = {
    "v1_2D": (0, np.float32, "P2"), "v2_2D": (1, np.float32, "P2"), "v3_2D": (2, np.float32, "P2"),
    "area2": (3, np.float32, "P"), "normal_view": (4, np.float32, "P3"), "v_depth": (5, np.float32, "P3"),
    "depth": (6, np.float32, "P"), "rgb": (7, np.float32, "P3"), "clamped": (8, np.uint8, "P3"),
    "point_offsets": (9, np.uint32, "P"), "tiles_touched": (10, np.uint32, "P"),
    "rect_min": (11, np.uint32, "P2"), "rect_max": (12, np.uint32, "P2"),
    "keys_unsorted": (13, np.uint64, "N"), "keys": (14, np.uint64, "N"),
    "vals_unsorted": (15, np.uint32, "N"), "vals": (16, np.uint32, "N"),
    "ranges": (17, np.uint32, "T2"), "n_contrib": (18, np.uint32, "HW"), "final_T": (19, np.float32, "HW"),
    "v1_view": (20, np.float32, "P3"), "v2_view": (21, np.float32, "P3"), "v3_view": (22, np.float32, "P3"),
}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "ts2d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libts2d_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_void_p
        L.ts2d_oracle_forward.restype = C.c_int
        L.ts2d_oracle_forward.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, fp, fp, fp, fp, fp,
                                          C.c_int, C.c_int, fp, ip, fp, fp, fp, fp, C.c_int, C.POINTER(vp)]
        L.ts2d_oracle_backward.restype = C.c_int
        L.ts2d_oracle_backward.argtypes = [vp, C.c_float, C.c_float, fp, fp, fp, C.c_int, C.c_int, C.c_int,
                                           C.c_float, C.c_float, fp, fp, fp, fp, fp, ip, fp, fp, fp, fp, fp, fp,
                                           fp, fp]
        L.ts2d_oracle_free.argtypes = [vp]
        L.ts2d_oracle_num_rendered.restype = C.c_int64
        L.ts2d_oracle_num_rendered.argtypes = [vp]
        L.ts2d_oracle_grid_x.argtypes = [vp]
        L.ts2d_oracle_grid_y.argtypes = [vp]
        L.ts2d_oracle_field.restype = vp
        L.ts2d_oracle_field.argtypes = [vp, C.c_int]
        L.ts2d_oracle_higher_msb.restype = C.c_uint32
        L.ts2d_oracle_higher_msb.argtypes = [C.c_uint32]
        L.ts2d_oracle_num_threads.restype = C.c_int
        L.ts2d_oracle_sh_color.argtypes = [C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, C.POINTER(C.c_uint8)]
        L.ts2d_oracle_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _fp(a: Optional[np.ndarray]):
    if a is None or a.size == 0:
        return C.cast(None, C.POINTER(C.c_float))
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleState:
    """Owns the oracle-side equivalents of geometryBuffer / binningBuffer / imageBuffer."""

    def __init__(self, handle, P, W, H, rich_info):
        self._h = handle
        self.P, self.W, self.H, self.rich_info = P, W, H, rich_info

    @property
    def num_rendered(self) -> int:
        return int(lib().ts2d_oracle_num_rendered(self._h))

    @property
    def grid(self):
        return lib().ts2d_oracle_grid_x(self._h), lib().ts2d_oracle_grid_y(self._h)

    def field(self, name: str) -> np.ndarray:
        idx, dt, kind = _FIELDS[name]
        gx, gy = self.grid
        n = {"P": self.P, "P2": 2 * self.P, "P3": 3 * self.P, "N": self.num_rendered, "T2": 2 * gx * gy,
             "HW": self.W * self.H}[kind]
        ptr = lib().ts2d_oracle_field(self._h, idx)
        if n == 0 or not ptr:
            out = np.zeros(0, dtype=dt)
        else:
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            out = np.frombuffer(buf, dtype=dt).copy()
        if kind in ("P2", "T2"):
            out = out.reshape(-1, 2)
        elif kind == "P3":
            out = out.reshape(-1, 3)
        elif kind == "HW":
            out = out.reshape(self.H, self.W)
        return out

    def close(self):
        if self._h:
            lib().ts2d_oracle_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rasterize_triangles(image_width, image_height, tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree,
                        gamma, scale_modifier, background_depth, background, vertex, shs, feature, opacity,
                        back_culling, rich_info, debug=False, variant=2):
    """Oracle counterpart of `_C.rasterize_triangles` (R2D/src/extension_interface.cu:19-152).

    Returns (num_rendered, out_feature, radii, depth, normal, contrib_sum, contrib_max, state)."""
    vertex = _f32(vertex)
    if vertex.ndim != 3 or vertex.shape[1:] != (3, 3):
        raise RuntimeError("vertex must have dimensions (num_points, 3, 3)")
    P = vertex.shape[0]
    shs = _f32(shs) if shs is not None else np.zeros(0, np.float32)
    feature = _f32(feature) if feature is not None else np.zeros(0, np.float32)
    # extension_interface.cu:44
    use_shs = feature.ndim <= 1 or (feature.shape[0] == 0 and shs.shape[0] > 0)
    if not use_shs and feature.ndim != 2:
        raise RuntimeError("feature must have dimensions (num_points, num_channels)")
    if use_shs and shs.ndim != 3:
        raise RuntimeError("shs must have dimensions (num_points, (1 + sh_degree) ** 2, 3)")
    Cn = 3 if use_shs else feature.shape[1]
    M = shs.shape[1] if (shs.ndim >= 2 and shs.shape[0] != 0) else 0
    if Cn > 3:
        raise RuntimeError("feature's num_channels can't be larger than MAX_CHANNELS")
    background = _f32(background).reshape(-1)
    if background.shape[0] != Cn:
        raise RuntimeError("background must have the same number of channels as feature")
    if gamma < 0:
        raise RuntimeError("gamma must be larger than 0")
    W, H = int(image_width), int(image_height)
    view, proj, cam = _f32(viewmatrix, (16,)), _f32(projmatrix, (16,)), _f32(campos, (3,))
    opacity = _f32(opacity).reshape(-1)
    out_feature = np.zeros((Cn, H, W), np.float32)
    radii = np.zeros(P, np.int32)
    if rich_info:
        depth = np.zeros((H, W), np.float32)
        normal = np.zeros((3, H, W), np.float32)
        csum = np.zeros(P, np.float32)
        cmax = np.zeros(P, np.float32)
    else:
        depth = normal = csum = cmax = np.zeros(0, np.float32)
    h = C.c_void_p()
    rc = lib().ts2d_oracle_forward(W, H, float(tan_fovx), float(tan_fovy), _fp(view), _fp(proj), _fp(cam), P,
                                   int(sh_degree), M, Cn, int(use_shs), float(gamma), float(background_depth),
                                   _fp(background), _fp(vertex), _fp(shs), _fp(feature), _fp(opacity),
                                   int(bool(back_culling)), int(bool(rich_info)), _fp(out_feature),
                                   radii.ctypes.data_as(C.POINTER(C.c_int)), _fp(depth), _fp(normal), _fp(csum),
                                   _fp(cmax), int(variant), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"ts2d_oracle_forward failed with code {rc}")
    st = OracleState(h, P, W, H, bool(rich_info))
    st._inputs = dict(use_shs=use_shs, M=M, C=Cn)
    return st.num_rendered, out_feature, radii, depth, normal, csum, cmax, st


def rasterize_triangles_backward(tan_fovx, tan_fovy, viewmatrix, projmatrix, campos, sh_degree, gamma,
                                 scale_modifier, background_depth, background, vertex, shs, feature, opacity,
                                 radii, state: OracleState, dL_dout_feature, dL_dout_depth, dL_dout_normal,
                                 rich_info, debug=False):
    """Oracle counterpart of `_C.rasterize_triangles_backward` (R2D/src/extension_interface.cu:154-260).

    Returns (dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity)."""
    vertex = _f32(vertex)
    P = vertex.shape[0]
    use_shs, M, Cn = state._inputs["use_shs"], state._inputs["M"], state._inputs["C"]
    shs = _f32(shs) if shs is not None else np.zeros(0, np.float32)
    feature = _f32(feature) if feature is not None else np.zeros(0, np.float32)
    view, proj, cam = _f32(viewmatrix, (16,)), _f32(projmatrix, (16,)), _f32(campos, (3,))
    background = _f32(background).reshape(-1)
    opacity = _f32(opacity).reshape(-1)
    radii = np.ascontiguousarray(radii, dtype=np.int32)
    g_feat = _f32(dL_dout_feature)
    g_depth = _f32(dL_dout_depth) if rich_info else np.zeros(0, np.float32)
    g_norm = _f32(dL_dout_normal) if rich_info else np.zeros(0, np.float32)
    dv = np.zeros((P, 3, 3), np.float32)
    dc = np.zeros((P, 2), np.float32)
    dsh = np.zeros((P, M, 3), np.float32)
    df = np.zeros((P, Cn), np.float32)
    dop = np.zeros((P, 1), np.float32)
    rc = lib().ts2d_oracle_backward(state._h, float(tan_fovx), float(tan_fovy), _fp(view), _fp(proj), _fp(cam),
                                    int(sh_degree), M, int(use_shs), float(gamma), float(background_depth),
                                    _fp(background), _fp(vertex), _fp(shs), _fp(feature), _fp(opacity),
                                    radii.ctypes.data_as(C.POINTER(C.c_int)), _fp(g_feat), _fp(g_depth),
                                    _fp(g_norm), _fp(dv), _fp(dc), _fp(dsh), _fp(df), _fp(dop))
    if rc != 0:
        raise RuntimeError(f"ts2d_oracle_backward failed with code {rc}")
    return dv, dc, dsh, df, dop


def sh_color(deg: int, shs: np.ndarray, pos: np.ndarray, campos: np.ndarray):
    """SH colour of n points (shs (n,M,3), pos (n,3)) seen from campos; returns (rgb (n,3), clamped (n,3) bool)."""
    shs, pos, campos = _f32(shs), _f32(pos), _f32(campos, (3,))
    n, M = shs.shape[0], shs.shape[1]
    rgb = np.zeros((n, 3), np.float32)
    cl = np.zeros((n, 3), np.uint8)
    lib().ts2d_oracle_sh_color(n, int(deg), M, _fp(shs), _fp(pos), _fp(campos), _fp(rgb),
                               cl.ctypes.data_as(C.POINTER(C.c_uint8)))
    return rgb, cl.astype(bool)


def higher_msb(n: int) -> int:
    return int(lib().ts2d_oracle_higher_msb(n))


def num_threads() -> int:
    return int(lib().ts2d_oracle_num_threads())


def set_num_threads(n: int) -> None:
    lib().ts2d_oracle_set_num_threads(int(n))
