"""GPU tests of the factored SH-gradient exchange (csrc/shgrad.hip, TS2D_FLAG_SH_FACTORED): the dense dL_dshs rebuilt
from per-view colour gradients must equal what the dense backward writes, up to fp32 summation order (the colour gradients of two
backward runs already differ by the order of the blend kernel's atomics)."""
import numpy as np
import pytest

import helpers
import synthetic

pytestmark = pytest.mark.gpu


def _views(s, n):
    """n cameras looking at the same triangles: the canonical one shifted sideways (same recipe as bench.py)."""
    cams = []
    for r in range(n):
        cam = synthetic.camera(s["image_width"], s["image_height"])
        if r > 0:
            view = cam["viewmatrix"].copy()
            shift = np.array([7.0 * r, -3.0 * r, 0.0], np.float32)
            view[3, :3] -= shift * np.array([-1, 1, -1], np.float32)
            cam["viewmatrix"] = view
            cam["projmatrix"] = (view @ synthetic.projection_matrix(cam["tanfovx"], cam["tanfovy"]).T).astype(np.float32)
            cam["campos"] = np.array([0, 0, synthetic.CAM_DIST], np.float32) + shift
        cams.append(cam)
    return cams


@pytest.mark.parametrize("variant", [2, 3])
@pytest.mark.parametrize("D,M", [(3, 16), (1, 16), (0, 1), (2, 9)])
def test_factored_equals_dense(variant, D, M):
    import torch
    from diff_triangle_rasterization_2D import parallel
    if variant == 3:
        from diff_triangle_rasterization_3D import TriangleRasterizer
    else:
        from diff_triangle_rasterization_2D import TriangleRasterizer

    P, W, H, V = 4000, 160, 128, 3
    s = synthetic.scene(P, W, H, D, seed=77, max_degree=int(round(M ** 0.5)) - 1)
    assert s["shs"].shape[1] == M
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    vertex = t(s["vertex"]).requires_grad_(True)
    opacity = t(s["opacity"]).requires_grad_(True)
    shs = t(s["shs"]).requires_grad_(True)
    g = [t(s["dL_dout_feature"]), t(s["dL_dout_depth"]), t(s["dL_dout_normal"])]

    def run(cam):
        sc = dict(s, **cam)
        rs = helpers.hip_settings(sc)
        c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
        out = TriangleRasterizer(rs)(vertex, c2d, opacity, shs=shs)
        torch.autograd.backward([out[0], out[2], out[3]], g)

    cams = _views(s, V)
    dense = []
    for cam in cams:
        shs.grad = None
        run(cam)
        dense.append(shs.grad.clone())
    vgrad_dense = vertex.grad.clone()

    shs.grad = None
    vertex.grad = None
    with parallel.factored_sh_grads() as sink:
        for cam in cams:
            run(cam)
    assert shs.grad is None and len(sink.colors) == V
    # everything else is untouched by the factored mode (equal up to the order of the blend kernel's fp32 atomics)
    assert helpers.rel_l2(vertex.grad.cpu().numpy(), vgrad_dense.cpu().numpy()) < 1e-5
    one = parallel.ShGradSink()
    one.append(sink.colors[0], sink.campos[0])
    first = parallel.exchange_factored_sh_grads(one, vertex, D, M)
    # one view: same expressions as the dense backward; the inputs differ only by the atomics order of two runs
    assert helpers.rel_l2(first.cpu().numpy(), dense[0].cpu().numpy()) < 1e-5
    total = parallel.exchange_factored_sh_grads(sink, vertex, D, M)
    want = sum(d.double() for d in dense)
    assert helpers.rel_l2(total.cpu().numpy(), want.cpu().numpy()) < 1e-5
    assert (total[:, (D + 1) ** 2:] == 0).all()
