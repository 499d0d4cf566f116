"""Runs scenes through the reference's own kernels (oracle/_ref) in a SEPARATE process and saves the outputs: the reference aborts on
some degenerate inputs (it was never exercised on them), and an abort must not take the test session down with it.

    python tests/ref_worker.py <seed> <out.npz>       one configuration of tests/test_fuzz_gpu.py:_case through the default build
    python tests/ref_worker.py --serve                request loop (tests/helpers.py:ref3d_builds): one line per request on stdin,
                                                      "<scene.npz> <out.npz> <build>[,<build>...]"; answers "ok" / "error <text>" on stdout
"""
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "triangle-splatting_amd"),
                os.path.dirname(os.path.abspath(__file__))]
import numpy as np  # noqa: E402

import ref_build  # noqa: E402


def load_scene(path):
    z = np.load(path, allow_pickle=False)
    s = {}
    for k in z.files:
        v = z[k]
        s[k] = v.item() if v.ndim == 0 else v
    meta = {k: s.pop(k) for k in ("__rich", "__back", "__use_feature", "__variant")}
    return s, bool(meta["__rich"]), bool(meta["__back"]), bool(meta["__use_feature"]), int(meta["__variant"])


if sys.argv[1] == "--serve":
    print("ready", flush=True)
    for line in sys.stdin:
        parts = line.split()
        if not parts:
            continue
        try:
            s, rich, back, use_feature, variant = load_scene(parts[0])
            res = {}
            for b in parts[2].split(","):
                rf = ref_build.forward_backward(s, rich, back, use_feature=use_feature, variant=variant, build=b)
                res.update({f"{b}/{k}": np.asarray(v) for k, v in rf.items()})
            np.savez(parts[1], **res)
            print("ok", flush=True)
        except Exception as e:  # a Python-level failure is reported, not swallowed
            print("error " + repr(e).replace("\n", " "), flush=True)
else:
    import test_fuzz_gpu as F
    seed, out = int(sys.argv[1]), sys.argv[2]
    s, variant, rich, back, use_feature = F._case(seed)
    rf = ref_build.forward_backward(s, rich, back, use_feature=use_feature, variant=variant)
    np.savez(out, **{k: np.asarray(v) for k, v in rf.items()})
