"""Synthetic random-triangle scenes S(P, W, H, D, seed) used by tests and bench.py.

Recipe: SURVEY.md section 8(d) / BASELINE.md section 4, which generalise the input recipe of the
reference's smoke executable (R2D/main.cu:10-37,77-79): fixed camera at (0,0,d) looking down -z,
`viewmatrix`/`projmatrix` in the reference's transposed (row-vector) convention
(src/diff_recon/utils/camera.py:112-115), triangle centroids uniform in the frustum-filling box,
vertices = centroid + N(0, sigma^2 I), uniform opacity / SH coefficients / upstream gradients.

Pure numpy (no torch, no GPU) so the same arrays can feed the oracle and the HIP path.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

TAN_FOVX = 0.3148  # R2D/main.cu:12
CAM_DIST = 1200.0  # R2D/main.cu:17,22
ZNEAR, ZFAR = 1.0, 1000.0  # src/diff_recon/utils/camera.py:109-110


def projection_matrix(tan_fovx: float, tan_fovy: float, znear: float = ZNEAR, zfar: float = ZFAR) -> np.ndarray:
    """Same matrix as the reference's getProjectionMatrix (camera.py:15-35), returned un-transposed."""
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 1.0 / tan_fovx
    P[1, 1] = 1.0 / tan_fovy
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera(W: int, H: int, tan_fovx: float = TAN_FOVX, dist: float = CAM_DIST) -> Dict[str, object]:
    tan_fovy = tan_fovx * H / W
    view = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, dist, 1]], np.float32)  # main.cu:14-17
    proj = (view @ projection_matrix(tan_fovx, tan_fovy).T).astype(np.float32)  # camera.py:113-114
    campos = np.array([0, 0, dist], np.float32)
    return dict(image_width=W, image_height=H, tanfovx=float(tan_fovx), tanfovy=float(tan_fovy),
                viewmatrix=view, projmatrix=proj, campos=campos)


def scene(P: int, W: int, H: int, D: int = 3, seed: int = 42, edge_px: float = 6.0, mode: str = "frustum",
          max_degree: int | None = None, with_grads: bool = True) -> Dict[str, object]:
    """Returns a dict with the camera fields plus vertex (P,3,3), shs (P,M,3), opacity (P,1),
    background (3,), background_depth, gamma, sh_degree and (optionally) upstream gradients.

    mode="frustum": centroids fill the view frustum at z in [0,200], sigma chosen for a mean projected
                    edge of `edge_px` pixels.
    mode="maincu":  every vertex uniform in the whole box of R2D/main.cu:29 (huge triangles; stress case).
    mode="centered": like "frustum" but the centroids are concentrated about the optical axis (an object in the middle of the image,
                    empty borders): the load is far from uniform over the tiles.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = camera(W, H)
    tx, ty = cam["tanfovx"], cam["tanfovy"]
    M = ((max_degree if max_degree is not None else D) + 1) ** 2
    if mode == "maincu":
        scale = np.array([1200.0, 600.0, 200.0], np.float32)
        shift = np.array([600.0, 300.0, 0.0], np.float32)
        vertex = rng.random((P, 3, 3), dtype=np.float32) * scale - shift
    else:
        z = rng.random((P, 1), dtype=np.float32) * 200.0
        zv = CAM_DIST - z  # view-space depth of the centroid
        if mode == "centered":
            # an object-centric view (NeRF-synthetic-like): centroids normally distributed about the optical axis, sigma = 0.2 of the half
            # extent, clipped to the frustum -- the middle of the image carries most of the work, the borders almost none
            ux = np.clip(rng.standard_normal((P, 1), dtype=np.float32) * np.float32(0.2), -1, 1)
            uy = np.clip(rng.standard_normal((P, 1), dtype=np.float32) * np.float32(0.2), -1, 1)
        else:
            ux = rng.random((P, 1), dtype=np.float32) * 2 - 1
            uy = rng.random((P, 1), dtype=np.float32) * 2 - 1
        cx = ux * (zv * tx)
        cy = uy * (zv * ty)
        centroid = np.concatenate([cx, cy, z], axis=1)[:, None, :]
        px_per_unit = 0.5 * W / ((CAM_DIST - 100.0) * tx)
        sigma = edge_px / (math.sqrt(math.pi) * px_per_unit)  # E|edge_2D| = sqrt(pi) * sigma for N(0, 2 sigma^2 I_2)
        vertex = centroid + rng.standard_normal((P, 3, 3), dtype=np.float32) * np.float32(sigma)
    out = dict(cam)
    out.update(
        vertex=np.ascontiguousarray(vertex, np.float32),
        shs=rng.random((P, M, 3), dtype=np.float32),
        opacity=rng.random((P, 1), dtype=np.float32),
        background=np.zeros(3, np.float32),
        background_depth=5000.0, gamma=1.0, scale_modifier=1.0, sh_degree=D,
    )
    if with_grads:
        out.update(
            dL_dout_feature=rng.random((3, H, W), dtype=np.float32),
            dL_dout_depth=rng.random((H, W), dtype=np.float32),
            dL_dout_normal=rng.random((3, H, W), dtype=np.float32),
        )
    return out
