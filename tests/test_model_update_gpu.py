"""GPU test of the fused densification statistics (csrc/model_update.hip through diff_recon_hip.DensificationStats) against
the reference's six boolean-mask statements (src/diff_recon/models/VanillaTS_model.py:355-363) restated with eager torch
indexing -- the same torch operations, applied to the same tensors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eager_update(state, radii, c2d_grad, csum, cmax):
    import torch
    vis = radii > 0
    state["gradient_accum"][vis] += torch.norm(c2d_grad[vis, :2], dim=-1)
    state["gradient_denom"][vis] += 1
    state["contrib_sum"][vis] = torch.max(state["contrib_sum"][vis], csum[vis])
    state["contrib_max"][vis] = torch.max(state["contrib_max"][vis], cmax[vis])
    state["contrib_denom"][vis] += 1
    state["max_radii2D"][vis] = torch.max(state["max_radii2D"][vis], radii[vis])


def test_statistics_match_reference_statements():
    import torch
    from diff_recon_hip import DensificationStats

    P = 100_003
    g = torch.Generator(device="cuda").manual_seed(5)
    stats = DensificationStats(P, "cuda")
    ref = {k: torch.zeros(P, device="cuda") for k in ("gradient_accum", "gradient_denom", "max_radii2D", "contrib_sum", "contrib_max",
                                                     "contrib_denom")}
    for it in range(4):
        radii = torch.randint(-1, 40, (P,), device="cuda", generator=g, dtype=torch.int32).clamp(min=0)
        c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
        c2d.grad = torch.randn((P, 2), device="cuda", generator=g)
        csum = torch.rand(P, device="cuda", generator=g) * 3
        cmax = torch.rand(P, device="cuda", generator=g)
        stats.update({"radii": radii, "center2D": c2d, "contrib_sum": csum, "contrib_max": cmax})
        _eager_update(ref, radii, c2d.grad, csum, cmax)
    for k, v in ref.items():
        assert torch.allclose(getattr(stats, k), v, rtol=1e-6, atol=0), k
    assert (stats.gradient_denom == stats.contrib_denom).all() and stats.gradient_denom.max() <= 4

    # without rich_info only the gradient / radius statistics move
    before = stats.contrib_sum.clone()
    c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
    c2d.grad = torch.ones((P, 2), device="cuda")
    stats.update({"radii": torch.ones(P, device="cuda", dtype=torch.int32), "center2D": c2d})
    assert torch.equal(stats.contrib_sum, before) and stats.gradient_denom.min() >= 1

    # prune / grow keep the six arrays aligned (VanillaTS_model.py:228-235, 309-315)
    mask = torch.rand(P, device="cuda", generator=g) < 0.3
    kept = stats.gradient_accum[~mask].clone()
    stats.prune(mask)
    stats.grow(17)
    assert len(stats) == int((~mask).sum()) + 17 and torch.equal(stats.gradient_accum[:-17], kept)
    assert (stats.max_radii2D[-17:] == 0).all()
    with pytest.raises(RuntimeError):
        stats.update({"radii": torch.ones(len(stats), device="cuda", dtype=torch.int32),
                      "center2D": torch.zeros((len(stats), 2), device="cuda")})


def test_multi_view_update_equals_sequential_updates():
    """num_views > 1 (the all-gathered inputs of image-parallel training) == applying the views one after another."""
    import ctypes as C
    import torch
    from diff_recon_hip import DensificationStats
    from diff_recon_hip.model_update import _lib

    P, V = 5000, 3
    g = torch.Generator(device="cuda").manual_seed(9)
    radii = torch.randint(0, 3, (V, P), device="cuda", generator=g, dtype=torch.int32)
    grad = torch.randn((V, P, 2), device="cuda", generator=g)
    csum, cmax = torch.rand((V, P), device="cuda", generator=g), torch.rand((V, P), device="cuda", generator=g)
    a, b = DensificationStats(P, "cuda"), DensificationStats(P, "cuda")
    for v in range(V):
        c2d = torch.zeros((P, 2), device="cuda", requires_grad=True)
        c2d.grad = grad[v].clone()
        a.update({"radii": radii[v], "center2D": c2d, "contrib_sum": csum[v], "contrib_max": cmax[v]})
    rc = _lib.tsm_training_statistic(P, V, radii.data_ptr(), grad.data_ptr(), csum.data_ptr(), cmax.data_ptr(),
                                     b.gradient_accum.data_ptr(), b.gradient_denom.data_ptr(), b.max_radii2D.data_ptr(),
                                     b.contrib_sum.data_ptr(), b.contrib_max.data_ptr(), b.contrib_denom.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for k in ("gradient_accum", "gradient_denom", "max_radii2D", "contrib_sum", "contrib_max", "contrib_denom"):
        assert torch.allclose(getattr(a, k), getattr(b, k), rtol=1e-6), k
