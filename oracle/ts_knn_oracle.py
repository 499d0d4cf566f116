"""CPU oracle of the nearest-neighbour helpers -- TEST INFRASTRUCTURE ONLY.

Both functions of submodules/simple-knn are exact searches (simple_knn.cu:153-235 prune boxes but never approximate), so
the oracle is the definition itself, evaluated with scipy's exact k-d tree in float64:
    mean_dist3(points)[i]      = mean of the 3 smallest |p_i - p_j|^2, j != i          (simple_knn.cu:153-187)
    nearest_other(points, g)[i] = argmin_j |p_i - p_j|^2 over j // g != i // g           (simple_knn.cu:189-235)
Parity status: the reference ships no expected outputs (main.cu:50-75 only prints); its extension built for gfx950
(oracle/build_ref.py -> oracle/_ref/_refknn_C.so) is compared with the product on the GPU (tests/test_reference_gpu.py:
distances to 2e-6, indices exactly), which pins the tie rule that this float64 oracle cannot.
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import cKDTree

FLT_MAX = float(np.finfo(np.float32).max)


def mean_dist3(points: np.ndarray) -> np.ndarray:
    p = np.asarray(points, np.float64)
    n = len(p)
    if n == 0:
        return np.zeros((0,), np.float64)
    k = min(4, n)
    d, _ = cKDTree(p).query(p, k=k)
    d = np.atleast_2d(d).reshape(n, k)[:, 1:] ** 2  # drop the point itself
    if k < 4:  # missing neighbours count as FLT_MAX in float32 arithmetic (simple_knn.cu:160,186)
        pad = np.full((n, 4 - k), FLT_MAX)
        d = np.concatenate([d, pad], 1)
        with np.errstate(over="ignore"):
            return ((d[:, 0].astype(np.float32) + d[:, 1].astype(np.float32) + d[:, 2].astype(np.float32)) / np.float32(3)).astype(np.float64)
    return d.sum(1) / 3.0


def nearest_other(points: np.ndarray, group: int):
    """Returns (index, squared distance) of the nearest point outside the own group; brute force per group chunk."""
    p = np.asarray(points, np.float64)
    n = len(p)
    tree = cKDTree(p)
    k = min(n, group + 1)
    idx = np.full(n, -1, np.int64)
    d2 = np.full(n, np.inf)
    dd, ii = tree.query(p, k=k)
    dd, ii = np.atleast_2d(dd).reshape(n, k), np.atleast_2d(ii).reshape(n, k)
    own = (ii // group) == (np.arange(n)[:, None] // group)
    for i in range(n):
        ok = np.nonzero(~own[i])[0]
        if len(ok):
            idx[i], d2[i] = ii[i, ok[0]], dd[i, ok[0]] ** 2
    return idx, d2
