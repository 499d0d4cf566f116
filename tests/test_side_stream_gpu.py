"""Round 6: the SH colours of the per-triangle kernel on a library-owned side stream beside the ordering chain -- built, measured
at the headline (profiles/r06_side_stream.txt) and NOT adopted: it lives in the lab library only.  What stays pinned here: the split kernels
(per-triangle kernel without colours + colour kernel) produce exactly the single launch's state, and the caller-side `center2D_sink`."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
LAB_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "bin", "libts2d_lab.so")


def test_side_stream_and_single_launch_leave_identical_state_and_outputs():
    if not os.path.exists(LAB_LIB):
        pytest.skip("tools/bin/libts2d_lab.so not built")
    e = dict(os.environ, TS2D_LIBRARY_PATH=LAB_LIB, LAB_SIDE_STREAM="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lab_worker.py")], env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("LAB_RESULT ")][-1][len("LAB_RESULT "):])
    assert len(res) == 4
    for case in res:
        for k, v in case.items():
            if k.startswith("int_"):
                assert v == 0.0, (case["P"], case["variant"], k)
            elif k not in ("P", "variant"):
                assert v < 2e-5, (case["P"], case["variant"], k, v)  # fp32 summation order of the atomics only


def test_center2D_sink_is_a_fresh_leaf_over_cached_zeros():
    from diff_triangle_rasterization_2D import center2D_sink
    a, b = center2D_sink(1000, "cuda"), center2D_sink(1000, "cuda")
    assert a.is_leaf and b.is_leaf and a.requires_grad and a is not b and a.grad is None
    assert a.data_ptr() == b.data_ptr() and float(a.abs().sum()) == 0.0 and a.shape == (1000, 2)
    a.backward(torch.ones_like(a))
    assert b.grad is None and float(a.grad.sum()) == 2000.0
