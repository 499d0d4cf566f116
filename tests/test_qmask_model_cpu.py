"""The quadrant mask of csrc/ts2d_support.h, restated in numpy fp32 (tools/sim/qmask_model.py), must contain every quadrant in which the blend
kernels' per-pixel test accepts a pixel -- random triangles from sub-pixel slivers to image-sized ones, opacities around the 1/255 threshold,
window exponents from 0.5 to 50, every tile around them.  (The kernels themselves are pinned on the GPU: tests/test_parity_gpu.py compares the
integer state and the images with the masks on; this test pins the FORMULA and its rounding margins, which no finite scene exercises fully.)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_quadrant_mask_never_misses_a_hit(seed, capsys, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tools", "sim"))
    import qmask_model
    monkeypatch.setattr(sys, "argv", ["qmask_model.py", "2500", str(seed)])
    assert qmask_model.main() == 0
    out = capsys.readouterr().out
    per_tile, affine = [l for l in out.splitlines() if "tightness" in l]
    for line in (per_tile, affine):   # quadrant_mask (per tile) and quad_anchor + quadrant_mask_affine (round 5: one setup per triangle, stepped
        assert "missed fp32 0, fp64 0" in line, line   # over its tile rectangle; the window here is up to 41 x 41 tiles for big triangles)
        tight = float(line.split("tightness ")[1].split(")")[0])
        assert tight > 0.9  # a mask that flags everything would also pass the line above


def test_quadrant_mask_of_the_3d_variant_never_misses_a_hit(capsys, monkeypatch):
    """3D variant (round 5): the mask is the 2D test on the view-space triangle scaled for the backward's G >= 1/255 test and PROJECTED (ts2d_support.h:
    quad_setup_3d), checked against the per-pixel ray / plane test of render3d_group.hip restated in fp32 and fp64 -- face-on to edge-on triangles,
    near to far, gamma 1 to 50.  Triangles whose plane's horizon crosses their tile rectangle (where the reference's arithmetic produces spurious
    hits with ecc = 1), with a scaled vertex near the camera, or with a projection thinner than 1e-3 px flag every quadrant."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "sim"))
    import qmask_model
    monkeypatch.setattr(sys, "argv", ["qmask_model.py", "1500", "21", "3d"])
    assert qmask_model.main() == 0
    out = capsys.readouterr().out
    per_tile, affine = [l for l in out.splitlines() if "tightness" in l]
    assert "missed fp32 0, fp64 0" in per_tile and affine.rstrip().endswith("missed 0"), out
    for line in (per_tile, affine):
        assert float(line.split("tightness ")[1].split(")")[0]) > 0.9
