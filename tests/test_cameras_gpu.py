"""GPU parity under arbitrary cameras: the six poses of tests/golden/camera.npz were built by the REFERENCE's Camera class
(random rotations / translations / fields of view / image sizes, incl. non-contiguous transposed matrices upstream), so
this covers view matrices other than the bench's axis-aligned one, for both rasterizer variants, including triangles
that straddle the image border, lie behind the camera, or are seen from behind (back-face culling on and off)."""
import os

import numpy as np
import pytest

import helpers
import test_parity3d_gpu as T3
import test_parity_gpu as T2

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cams():
    g = np.load(os.path.join(GOLD, "camera.npz"))
    n = len([k for k in g.files if k.startswith("W_")])
    return [{k.rsplit("_", 1)[0]: g[k] for k in g.files if k.endswith(f"_{i}")} for i in range(n)]


def _scene(c, P, seed, D=2):
    rng = np.random.default_rng(seed)
    W, H = int(c["W"]), int(c["H"])
    # keep the oracle fast: cap the image at ~0.2 Mpix by rescaling (tan_fov is what the kernels consume)
    sc = min(1.0, (200_000 / (W * H)) ** 0.5)
    W, H = max(16, int(W * sc)), max(16, int(H * sc))
    view = c["world_view_transform"].astype(np.float64)
    tx, ty = float(c["tan_fovx"]), float(c["tan_fovy"])
    z = rng.uniform(2.0, 40.0, P)
    z[: P // 20] = -rng.uniform(0.5, 10.0, P // 20)      # behind the camera
    x = rng.uniform(-1.15, 1.15, P) * np.abs(z) * tx    # some outside / straddling the frustum
    y = rng.uniform(-1.15, 1.15, P) * np.abs(z) * ty
    pw = np.stack([x, y, z, np.ones(P)], 1) @ np.linalg.inv(view)
    size = np.abs(z)[:, None, None] * tx * rng.uniform(4.0, 40.0, (P, 1, 1)) / W
    vertex = (pw[:, None, :3] + rng.normal(0, 1.0, (P, 3, 3)) * size).astype(np.float32)
    M = (D + 1) ** 2
    s = dict(image_width=W, image_height=H, tanfovx=tx, tanfovy=ty,
             viewmatrix=np.ascontiguousarray(c["world_view_transform"], np.float32),
             projmatrix=np.ascontiguousarray(c["full_proj_transform"], np.float32),
             campos=np.ascontiguousarray(c["camera_center"], np.float32), sh_degree=D, gamma=1.0, scale_modifier=1.0,
             background_depth=100.0, background=np.array([0.1, 0.0, 0.2], np.float32), vertex=vertex,
             shs=rng.uniform(0, 1, (P, M, 3)).astype(np.float32), opacity=rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32),
             dL_dout_feature=rng.uniform(0, 1, (3, H, W)).astype(np.float32),
             dL_dout_depth=rng.uniform(0, 1, (H, W)).astype(np.float32) * 0.01,
             dL_dout_normal=rng.uniform(0, 1, (3, H, W)).astype(np.float32))
    return s


@pytest.mark.parametrize("i", range(6))
@pytest.mark.parametrize("variant", [2, 3])
def test_reference_built_cameras(i, variant):
    c = _cams()[i]
    back_culling = bool(i % 2)
    s = _scene(c, 3000, seed=100 + i)
    of = helpers.oracle_forward(s, True, back_culling, variant=variant)
    ob = helpers.oracle_backward(s, of, True)
    hf = helpers.hip_forward_backward(s, True, back_culling, variant=variant)
    assert 0 < (of["radii"] > 0).sum() < 3000  # some culled (behind / outside / back-facing), most visible
    if variant == 2:
        T2._check_state(s, hf, of)
        T2._check_outputs(hf, of, ob, True)
    else:
        T3._check_state3d(s, hf, of)
        T3._check_outputs(s, hf, of, ob, True, back=back_culling)
