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
    assert "missed fp32 0, fp64 0" in out
    tight = float(out.split("tightness ")[1].split(")")[0])
    assert tight > 0.9  # a mask that flags everything would also pass the line above
