"""The quadrant masks on the GPU, with the real kernels: they only cull work that contributes nothing, so with every quadrant of every instance
flagged (the lab library's ts2d_lab_force_all_quadrants) each output must be what it is with the masks on -- images, depth, normals, radii and
n_contrib bit for bit (a pixel's blend order and arithmetic do not depend on what else its quadrant wave looked at), the atomically summed
outputs to the rounding of their summation order.  Scenes: both variants, slivers and heavy overdraw, triangles spanning many tiles (the affine
step over long rectangles), the main.cu recipe (the unstaged emission path), gamma 1 / 7 / 50, and -- for the 3D variant -- up to 30 % of the
triangles turned edge-on to their viewing ray, where the plane's horizon crosses the tile rectangle (csrc/ts2d_support.h: quad_setup_3d).
tools/sim/qmask_model.py pins the FORMULAS in numpy; this pins what runs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.path.join(ROOT, "tools", "bin", "libts2d_lab.so")


def test_the_masks_change_no_output():
    if not os.path.exists(LAB_LIB):
        pytest.skip("tools/bin/libts2d_lab.so not built")
    e = dict(os.environ, TS2D_LIBRARY_PATH=LAB_LIB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "qmask_worker.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("QMASK_RESULT ")][-1][len("QMASK_RESULT "):])
    assert len(res) == 9
    for case in res:
        assert case["all_quadrants_really_all"] and case["same_num_rendered"], case
        assert 0.05 < case["mask_bits_set_fraction"] < 0.95, case  # the masks do cull (and do not cull everything)
        for k in ("out_feature", "depth", "normal", "radii", "n_contrib"):
            assert case["exact_" + k], case
        for k in ("contrib_sum", "contrib_max", "dL_dvertex", "dL_dcenter2D", "dL_dshs", "dL_dopacity"):
            assert case[k] < 2e-6, case
