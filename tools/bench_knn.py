"""Times distCUDA2 / nearestNeighbor of the simple_knn drop-in on MI355X (init-time helpers; not the headline)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "triangle-splatting_amd")]
import torch
from simple_knn import distCUDA2, nearestNeighbor
res = []
for n in (99_999, 999_999, 3_000_000):
    g = torch.Generator(device="cuda").manual_seed(n)
    p = torch.rand((n, 3), device="cuda", generator=g) * 100
    for name, fn in (("distCUDA2", lambda: distCUDA2(p)), ("nearestNeighbor(bs=3)", lambda: nearestNeighbor(p, 3))):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        res.append({"fn": name, "points": n, "ms": round((time.perf_counter() - t) / 3 * 1e3, 3)})
print(json.dumps(res))
