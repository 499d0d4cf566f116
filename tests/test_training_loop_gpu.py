"""End-to-end check that the drop-in pieces compose into a working optimisation loop (examples/train_synthetic.py):
render_view -> HIP rasterizer (2D and 3D) -> fused photometric loss -> Adam -> DensificationStats."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


@pytest.mark.parametrize("rasterizer", ["2D", "3D"])
def test_loss_decreases(rasterizer):
    import torch
    import train_synthetic

    losses, stats, _ = train_synthetic.train(rasterizer, iters=60, triangles=4000, width=160, height=112, log=None)
    assert all(l == l for l in losses)  # finite
    assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])
    assert stats.gradient_denom.max() == 60 and stats.max_radii2D.max() > 0 and torch.isfinite(stats.gradient_accum).all()
