"""`diff_recon_hip.create_from_pcd` end to end on the HIP device (the neighbour search = simple_knn.distCUDA2, csrc/knn.hip) against the fixture the
reference's own method produced (tests/golden/create_from_pcd.npz; tests/test_model_init_cpu.py explains it), and as the first step of a training
loop: the triangles it makes render."""
import numpy as np
import pytest
import torch

from test_model_init_cpu import GOLD, case_kwargs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["plain", "twins_dup", "grid"])
def test_create_from_pcd_on_the_device_matches_the_reference(name):
    from diff_recon_hip import create_from_pcd
    g = np.load(GOLD, allow_pickle=False)
    torch.manual_seed(int(g[f"{name}/seed"]))
    normals = g[f"{name}/normals"]
    if not normals.any():
        # the reference drew its random normals with randn_like on ITS device (the CPU, when the fixture was made): take them from the same CPU
        # stream here; the offsets of the duplicated points follow on that stream in both (torch.rand on the CPU generator, then .to(device))
        normals = torch.randn((normals.shape[0], 3)).numpy()
    out = create_from_pcd(g[f"{name}/points"], g[f"{name}/colors"], normals, device="cuda", **case_kwargs(g, name))
    for key, ref in (("_vertex", "vertex"), ("_opacity", "opacity"), ("_f_dc", "f_dc"), ("_f_rest", "f_rest")):
        got, want = out[key].cpu().numpy(), g[f"{name}/{ref}"]
        assert got.shape == want.shape
        # the circum-radius is a root of a float32 sum of three squared distances on the device, of a float64 one in the fixture
        assert np.allclose(got, want, rtol=2e-5, atol=2e-5), (name, key, float(np.abs(got - want).max()))


def test_the_triangles_it_makes_render():
    """A point cloud on a sphere in front of the canonical camera -> create_from_pcd -> render_view: a non-trivial image, every triangle visible,
    gradients reach all four parameters (what configs[3]'s training starts from)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import synthetic
    import train_synthetic
    from diff_recon_hip import create_from_pcd, render_view
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(4)
    d = torch.randn((6000, 3), generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    pts = d * 40.0 + torch.tensor([0.0, 0.0, 100.0])  # a sphere of radius 40 inside the frustum of synthetic.camera (camera at z = 1200 looking down -z)
    cols = torch.rand((6000, 3), generator=g)
    params = create_from_pcd(pts, cols, d, max_sh_degree=1, init_opacity=0.5, back_culling=True)
    assert params["_vertex"].shape == (12000, 3, 3) and params["_f_rest"].shape == (12000, 3, 3)
    s = synthetic.scene(8, 320, 240, 1, seed=0)
    cam = train_synthetic.Camera(s, dev)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    pkg = render_view(cam, leaves["_vertex"], leaves["_f_dc"], leaves["_f_rest"], leaves["_opacity"], bg_color=torch.zeros(3, device=dev), gamma=1.0,
                      active_sh_degree=1, max_sh_degree=1, is_training=True, back_culling=True, rasterizer_type="3D")
    img = pkg["render"]
    assert float(img.max()) > 0.05 and int((pkg["radii"] > 0).sum()) > 3000  # about half of the twins face the camera
    img.sum().backward()
    for k, v in leaves.items():
        assert v.grad is not None and float(v.grad.abs().sum()) > 0, k
