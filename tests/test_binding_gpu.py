"""The drop-in boundary from the side the reference's own FFI binds: `bindings/_ts2d_torch_C.so` is a compiled torch C++
extension (pybind, the reference's `ext.cpp` signatures, R2D/ext.cpp:4-9 + R2D/src/extension_interface.h:7-62) that links
libts2d.so.  Three parity cases go through ITS `rasterize_triangles` / `rasterize_triangles_backward` and are checked against
the CPU oracle; the error path and the P = 0 path mirror the reference's."""
import importlib.util
import os

import numpy as np
import pytest

import helpers
import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "triangle-splatting_amd", "bindings", "_ts2d_torch_C.so")


def _ext():
    import torch  # noqa: F401  (libtorch before the extension)
    if not os.path.exists(SO):
        pytest.skip(f"{SO} not built (python triangle-splatting_amd/bindings/build_torch_ext.py)")
    spec = importlib.util.spec_from_file_location("_ts2d_torch_C", SO)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _run(ext, s, rich, use_feature=False, back=False):
    import torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    empty = torch.empty(0, device="cuda")
    shs = empty if use_feature else t(s["shs"])
    feature = t(s["feature"]) if use_feature else empty
    cam = (s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), int(s["sh_degree"]), float(s["gamma"]),
           float(s["scale_modifier"]), float(s["background_depth"]), t(s["background"]))
    vertex, opacity = t(s["vertex"]), t(s["opacity"])
    out = ext.rasterize_triangles(s["image_width"], s["image_height"], *cam, vertex, shs, feature, opacity, back, rich, False)
    n, img, radii, depth, normal, csum, cmax, gb, bb, ib = out
    gd = t(s["dL_dout_depth"]) if rich else empty
    gn = t(s["dL_dout_normal"]) if rich else empty
    bw = ext.rasterize_triangles_backward(*cam, vertex, shs, feature, opacity, n, radii, gb, bb, ib, t(s["dL_dout_feature"]), gd, gn, rich, False)
    res = dict(num_rendered=n, out_feature=img.cpu().numpy(), radii=radii.cpu().numpy())
    if rich:
        res.update(depth=depth.cpu().numpy(), normal=normal.cpu().numpy(), contrib_sum=csum.cpu().numpy(), contrib_max=cmax.cpu().numpy())
    dv, dc, dsh, df, dop = (x.cpu().numpy() for x in bw)
    res.update(dL_dvertex=dv, dL_dcenter2D=dc, dL_dopacity=dop, dL_dshs=dsh, dL_dfeature=df)
    return res


@pytest.mark.parametrize("P,W,H,D,rich,gamma,use_feature", [
    (10000, 256, 256, 0, True, 1.0, False),   # BASELINE.json configs[0]
    (6000, 200, 130, 3, True, 2.5, False),
    (4000, 97, 61, 0, False, 1.0, True),      # feature mode, rich_info off, ragged image
])
def test_parity_through_the_compiled_reference_side_binding(P, W, H, D, rich, gamma, use_feature):
    ext = _ext()
    s = synthetic.scene(P, W, H, D, seed=31 + P)
    s["gamma"] = gamma
    if use_feature:
        s["feature"] = np.random.default_rng(1).random((P, 3), dtype=np.float32)
        s["background"] = np.array([0.3, 0.1, 0.7], np.float32)
    hf = _run(ext, s, rich, use_feature)
    of = helpers.oracle_forward(s, rich, use_feature=use_feature)
    ob = helpers.oracle_backward(s, of, rich, use_feature=use_feature)
    assert hf["num_rendered"] == of["num_rendered"] and np.array_equal(hf["radii"], of["radii"])
    assert helpers.rel_l2(hf["out_feature"], of["out_feature"]) < 1e-4
    if rich:
        for k in ("depth", "normal", "contrib_sum", "contrib_max"):
            assert helpers.rel_l2(hf[k], of[k]) < 1e-4, k
    for k in ("dL_dvertex", "dL_dcenter2D", "dL_dopacity", "dL_dfeature" if use_feature else "dL_dshs"):
        assert helpers.rel_l2(hf[k], ob[k]) < 1e-3, k


def test_binding_checks_and_empty_input():
    import torch
    ext = _ext()
    s = synthetic.scene(50, 32, 32, 0, seed=1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    empty = torch.empty(0, device="cuda")
    cam = (s["tanfovx"], s["tanfovy"], t(s["viewmatrix"]), t(s["projmatrix"]), t(s["campos"]), 0, 1.0, 1.0, 10.0, t(s["background"]))
    with pytest.raises(RuntimeError, match="vertex must have dimensions"):
        ext.rasterize_triangles(32, 32, *cam, t(s["vertex"]).reshape(-1, 9), t(s["shs"]), empty, t(s["opacity"]), False, True, False)
    with pytest.raises(RuntimeError, match="gamma must be larger than 0"):
        bad = cam[:6] + (-1.0,) + cam[7:]
        ext.rasterize_triangles(32, 32, *bad, t(s["vertex"]), t(s["shs"]), empty, t(s["opacity"]), False, True, False)
    out = ext.rasterize_triangles(32, 32, *cam, torch.empty((0, 3, 3), device="cuda"), torch.empty((0, 1, 3), device="cuda"), empty,
                                  torch.empty((0, 1), device="cuda"), False, True, False)
    assert out[0] == 0 and float(out[1].abs().sum()) == 0.0 and out[7].numel() == 0  # extension_interface.cu:130


def test_the_package_runs_on_both_bindings():
    """Round 6: the compiled binding is the package's default, ctypes the fallback -- every test that goes through the package
    must be green on BOTH.  This process runs on the default (asserted); the structured parity cases, the speculative / sync-free forward tests
    (device-tensor background depth, capacity overflow, graph replay) and the exchange-bucket tests (preallocated outputs, factored SH gradients,
    ranged backward) run once more in a child process with TS2D_BINDING=ctypes."""
    import os
    import subprocess
    import sys
    from diff_triangle_rasterization_2D import _C
    assert _C.binding() == "compiled" or os.environ.get("TS2D_BINDING") == "ctypes" or os.environ.get("TS2D_LIBRARY_PATH")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, TS2D_BINDING="ctypes")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_parity_gpu.py"), os.path.join(here, "test_speculative_forward_gpu.py"), os.path.join(here, "test_async_forward_gpu.py"),
                        os.path.join(here, "test_factored_gpu.py"), os.path.join(here, "test_parity3d_gpu.py"),
                        "-k", "not full_size and not lab and not forced and not one_launch"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
