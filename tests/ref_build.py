"""Loader for the reference's own extensions built for gfx950 by oracle/build_ref.py (oracle/_ref/*.so; test infrastructure).
Exposes thin wrappers with the calling convention of tests/helpers.py so that the same scenes can go through the oracle, the
HIP product path and the REFERENCE's kernels."""
from __future__ import annotations

import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_cache = {}


def load(name: str):
    """name in {"_ref2d_C", "_ref2d_nofma_C", "_ref3d_C", "_ref3d_scalar_C", "_ref3d_nofma_C", "_refknn_C"}; skips the calling test when the build is absent."""
    if name not in _cache:
        path = os.path.join(ROOT, "oracle", "_ref", name + ".so")
        if not os.path.exists(path):
            pytest.skip(f"{path} not built (python oracle/build_ref.py in the build container)")
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _cache[name] = mod
    return _cache[name]


def forward_backward(s, rich_info=True, back_culling=False, use_feature=False, variant=2, device="cuda", build=None):
    """One scene through the reference's rasterize_triangles / rasterize_triangles_backward (R2D/ext.cpp:6-8, R3D/ext.cpp).
    `build` names another build of the same sources (oracle/build_ref.py CODEGEN_FLAGS), e.g. "_ref3d_scalar_C"."""
    import torch
    ref = load(build or ("_ref2d_C" if variant == 2 else "_ref3d_C"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    empty = torch.empty(0, device=device)
    shs = empty if use_feature else t(s["shs"])
    feature = t(s["feature"]) if use_feature else empty
    cam = (s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), int(s["sh_degree"]), float(s["gamma"]),
           float(s["scale_modifier"]), float(s["background_depth"]), t(s["background"]))
    vertex, opacity = t(s["vertex"]), t(s["opacity"])
    out = ref.rasterize_triangles(s["image_width"], s["image_height"], *cam, vertex, shs, feature, opacity, back_culling, rich_info, False)
    n, img, radii, depth, normal, csum, cmax, gb, bb, ib = out
    H, W = s["image_height"], s["image_width"]
    g_depth = t(s["dL_dout_depth"]) if rich_info else torch.empty(0, device=device)
    g_normal = t(s["dL_dout_normal"]) if rich_info else torch.empty(0, device=device)
    bw = ref.rasterize_triangles_backward(*cam, vertex, shs, feature, opacity, n, radii, gb, bb, ib, t(s["dL_dout_feature"]), g_depth, g_normal,
                                          rich_info, False)
    torch.cuda.synchronize()
    res = dict(num_rendered=int(n), out_feature=img.cpu().numpy(), radii=radii.cpu().numpy())
    if rich_info:
        res.update(depth=depth.cpu().numpy(), normal=normal.cpu().numpy(), contrib_sum=csum.cpu().numpy(), contrib_max=cmax.cpu().numpy())
    dv, dc, dsh, df, dop = (x.cpu().numpy() for x in bw)
    res.update(dL_dvertex=dv, dL_dcenter2D=dc, dL_dopacity=dop)
    res["dL_dfeature" if use_feature else "dL_dshs"] = df if use_feature else dsh
    return res


def _carve(buf, fields):
    """The reference's `obtain` (R2D/src/param_struct.h:11-17): every array starts at the next multiple of ALIGNMENT = 128 bytes
    (config.h:8) of the ADDRESS, in declaration order.  fields = [(name, numpy dtype, count)]; returns {name: numpy array} copied to the host."""
    base = buf.data_ptr()
    host = buf.cpu().numpy()
    out, p = {}, base
    for name, dt, count in fields:
        p = (p + 127) & ~127
        nbytes = np.dtype(dt).itemsize * count
        out[name] = host[p - base:p - base + nbytes].view(dt).copy()
        p += nbytes
    return out


def forward_integer_state(s, build="_ref2d_nofma_C", rich_info=True, back_culling=False, device="cuda", use_feature=False):
    """The integer / index state of ONE forward of the reference's 2D extension, read out of its three private buffers:
    num_rendered, radii, and -- decoded as the reference lays them out (GeometryState / BinningState / ImageState::fromChunk,
    R2D/src/param_struct.h:66-125) -- tiles_touched, point_offsets, the SORTED keys and instance list (point_list_keys, point_list) and the
    tile ranges (the reference allocates W*H of them and fills one per tile, rasterizer.cu:229-236)."""
    import torch
    ref = load(build)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    out = ref.rasterize_triangles(s["image_width"], s["image_height"], s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]),
                                  int(s["sh_degree"]), float(s["gamma"]), float(s["scale_modifier"]), float(s["background_depth"]), t(s["background"]),
                                  t(s["vertex"]), torch.empty(0, device=device) if use_feature else t(s["shs"]),
                                  t(s["feature"]) if use_feature else torch.empty(0, device=device), t(s["opacity"]), back_culling, rich_info, False)
    torch.cuda.synchronize()
    n, radii, gb, bb, ib = int(out[0]), out[2], out[7], out[8], out[9]
    P, W, H = s["vertex"].shape[0], s["image_width"], s["image_height"]
    T = ((W + 15) // 16) * ((H + 15) // 16)
    geo = _carve(gb, [("v1", np.float32, 2 * P), ("v2", np.float32, 2 * P), ("v3", np.float32, 2 * P), ("area2", np.float32, P),
                      ("normal_view", np.float32, 3 * P), ("v_depth", np.float32, 3 * P), ("depth", np.float32, P), ("rgb", np.float32, 3 * P),
                      ("clamped", np.uint8, 3 * P), ("point_offsets", np.uint32, P), ("tiles_touched", np.uint32, P)])
    res = dict(num_rendered=n, radii=radii.cpu().numpy(), tiles_touched=geo["tiles_touched"], point_offsets=geo["point_offsets"], depth=geo["depth"])
    if n > 0:
        b = _carve(bb, [("keys_unsorted", np.uint64, n), ("keys", np.uint64, n), ("list_unsorted", np.uint32, n), ("point_list", np.uint32, n)])
        res.update(keys=b["keys"], point_list=b["point_list"])
        res["ranges"] = _carve(ib, [("ranges", np.uint32, 2 * W * H)])["ranges"].reshape(-1, 2)[:T]
    return res
