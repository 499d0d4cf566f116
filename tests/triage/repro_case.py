"""Re-run one fuzz case (tests/test_fuzz_gpu.py::_case) through the product with per-stage checking: python repro_case.py SEED [debug]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, ".."), os.path.join(HERE, "..", ".."), os.path.join(HERE, "..", "..", "triangle-splatting_amd")]
import helpers
import test_fuzz_gpu as F

seed = int(sys.argv[1])
s, variant, rich, back, use_feature = F._case(seed)
print("case", seed, "variant", variant, "rich", rich, "back", back, "feature", use_feature, "P", s["vertex"].shape[0], s["image_width"], s["image_height"], flush=True)
hf = helpers.hip_forward_backward(s, rich, back, use_feature=use_feature, variant=variant, debug=len(sys.argv) > 2)
print("num_rendered", hf["num_rendered"], flush=True)
