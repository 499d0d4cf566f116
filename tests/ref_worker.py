"""Runs ONE random configuration (tests/test_fuzz_gpu.py:_case) through the reference's own kernels (oracle/_ref) in a separate
process and saves the outputs: the reference aborts on some degenerate inputs (it was never exercised on them), and an
abort must not take the test session down with it.    python tests/ref_worker.py <seed> <out.npz>"""
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "triangle-splatting_amd"),
                os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402

import ref_build  # noqa: E402
import test_fuzz_gpu as F  # noqa: E402

seed, out = int(sys.argv[1]), sys.argv[2]
s, variant, rich, back, use_feature = F._case(seed)
rf = ref_build.forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
np.savez(out, **{k: np.asarray(v) for k, v in rf.items()})
