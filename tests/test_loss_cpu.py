"""CPU tests of the photometric-loss oracle (oracle/ts_loss_oracle.py) against golden vectors produced by the
REFERENCE's own SSIM / L1 classes and torch autograd (tests/golden/photometric.npz, generator tests/golden/make_golden.py),
plus the host-side argument checks of the drop-in module."""
import os

import numpy as np
import pytest

from oracle import ts_loss_oracle as LO

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "photometric.npz"))


@pytest.mark.parametrize("i", range(int(GOLD["n"])))
def test_oracle_matches_reference_loss_and_autograd(i):
    img, gt = GOLD[f"img{i}"], GOLD[f"gt{i}"]
    w1, ws = GOLD[f"w{i}"]
    loss, l1, sl, grad = LO.photometric_loss(img, gt, w1, ws)
    assert abs(l1 - float(GOLD[f"l1_{i}"])) < 1e-12
    # the window is built in float32 (trainer_utils.py:17-29); torch's and numpy's fp32 exp differ in the last bit,
    # which bounds the agreement of everything downstream at ~1e-7
    assert abs(sl - float(GOLD[f"ssim_loss_{i}"])) < 5e-7
    assert abs(loss - float(GOLD[f"loss_{i}"])) < 5e-7
    g_ref = GOLD[f"grad_{i}"]
    assert np.abs(grad - g_ref).max() <= 1e-5 * np.abs(g_ref).max()
    # the reference's own float32 evaluation stays within fp32 noise of the float64 one
    assert abs(float(GOLD[f"loss_f32_{i}"]) - loss) < 1e-5
    assert np.linalg.norm(GOLD[f"grad_f32_{i}"] - g_ref) <= 1e-4 * np.linalg.norm(g_ref)


def test_window_is_zero_padded_and_normalised():
    k = LO.gaussian_kernel_2d()
    assert k.shape == (11, 11) and abs(k.sum() - 1.0) < 1e-6 and np.allclose(k, k.T)
    x = np.ones((1, 30, 30))
    w = LO.window(x)
    assert abs(w[0, 15, 15] - 1.0) < 1e-6          # interior: full window
    assert abs(w[0, 0, 0] - k[5:, 5:].astype(np.float64).sum()) < 1e-9  # corner: only the in-image quadrant contributes (zero padding)


def test_batched_input_folds_into_channels():
    rng = np.random.default_rng(0)
    a, b = rng.random((2, 3, 16, 18)), rng.random((2, 3, 16, 18))
    l4 = LO.photometric_loss(a, b, 0.8, 0.2)
    l3 = LO.photometric_loss(a.reshape(6, 16, 18), b.reshape(6, 16, 18), 0.8, 0.2)
    assert abs(l4[0] - l3[0]) < 1e-15 and np.array_equal(l4[3].reshape(6, 16, 18), l3[3])


# ---- host logic of diff_recon_hip.model_forward (pure torch helpers; the render itself is covered by the GPU suite) ----
def test_model_forward_helpers(hip_lib_built):
    import torch
    from diff_recon_hip import gamma_rescale_ratio, rescale_triangles, ste_opacity

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gamma_rescale.npz"))
    for gamma, ratio in zip(g["gamma"], g["ratio"]):  # fixture from the reference's formula evaluated with scipy
        assert abs(gamma_rescale_ratio(float(gamma)) - float(ratio)) < 1e-12
    v = torch.rand((5, 3, 3), dtype=torch.float64)
    r = rescale_triangles(v, 0.5)
    assert torch.allclose(r.mean(1), v.mean(1)) and torch.allclose(r - r.mean(1, keepdim=True), 0.5 * (v - v.mean(1, keepdim=True)))
    rt = rescale_triangles(v, torch.full((5,), 2.0, dtype=torch.float64))
    assert torch.allclose(rt - rt.mean(1, keepdim=True), 2.0 * (v - v.mean(1, keepdim=True)))
    o = torch.tensor([[0.2], [0.7]], requires_grad=True)
    s = ste_opacity(o, 0.5)
    assert s.tolist() == [[0.0], [1.0]]
    s.sum().backward()
    assert o.grad.tolist() == [[1.0], [1.0]]  # straight-through


def test_depth_normal_oracle_matches_the_reference_class():
    """oracle/ts_loss_oracle.py:depth_normal_loss against tests/golden/depth_normal.npz = the reference's DepthNormalLoss
    (trainer_utils.py:204-257) + torch autograd, executed by tests/golden/make_golden.py: scale factors 0.5 (every shipped config),
    0.25 and None, odd image sizes, a zero rendered normal (eps branch of F.normalize)."""
    import os
    import numpy as np
    from oracle import ts_loss_oracle as O
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_normal.npz"))
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for i, (H, W, s, q, tx, ty) in enumerate(z["cases"]):
        loss, dd, dn = O.depth_normal_loss(z[f"depth{i}"], z[f"normal{i}"], tx, ty, None if s < 0 else float(s), q)
        assert abs(loss - float(z[f"loss{i}"])) < 2e-6 * abs(float(z[f"loss{i}"]))
        assert rel(dd, z[f"ddepth{i}"]) < 2e-5 and rel(dn, z[f"dnormal{i}"]) < 2e-5


def test_aux_loss_oracles_match_the_reference_classes():
    """oracle/ts_loss_oracle.py: dog_mask / dog_loss / smoothness_mask / smoothness_loss against tests/golden/aux_losses.npz = the reference's
    DoGLoss and SmoothnessLoss (trainer_utils.py:105-148, 181-201) + torch autograd, executed by tests/golden/make_golden.py aux_losses: scale
    factors 0.5 (the default), 0.25, 0.7 (non-dyadic) and 1, freq on both sides of 50, one to three channels, pixels where image == target
    (sign(0) = 0).  The masks are threshold decisions of float32 pipelines: a pixel may differ from the float64
    restatement only where its value sits within 1e-5 of the threshold."""
    import os
    import numpy as np
    from oracle import ts_loss_oracle as O
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aux_losses.npz"))
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    for i, (C, H, W, s, freq, q) in enumerate(z["cases"]):
        s = None if s < 0 else float(s)
        img, gt = z[f"img{i}"], z[f"gt{i}"]
        aux = {}
        m = O.dog_mask(gt, int(freq), s, aux)
        differ = m != z[f"dog_mask{i}"]
        assert (np.abs(aux["normalized"][differ] - 0.5) < 1e-5).all() and differ.mean() < 2e-3, i
        loss, grad = O.dog_loss(img, gt, int(freq), s, mask=z[f"dog_mask{i}"])   # the losses on the reference's own mask
        assert abs(loss - float(z[f"dog_loss{i}"])) < 2e-6 * float(z[f"dog_loss{i}"]) and rel(grad, z[f"dog_grad{i}"]) < 1e-6, i
        aux = {}
        m = O.smoothness_mask(gt, float(q), s, aux)
        differ = m != z[f"smooth_mask{i}"]
        assert (np.abs(aux["U"][differ] - aux["threshold"]) < 1e-5).all() and differ.mean() < 2e-3, i
        loss, grad = O.smoothness_loss(img, gt, float(q), s, mask=z[f"smooth_mask{i}"])
        assert abs(loss - float(z[f"smooth_loss{i}"])) < 2e-6 * float(z[f"smooth_loss{i}"]) and rel(grad, z[f"smooth_grad{i}"]) < 2e-5, i
