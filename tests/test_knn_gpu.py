"""GPU parity of the simple_knn drop-in (csrc/knn.hip through simple_knn -> ctypes -> C ABI) against the exact-search
oracle (oracle/ts_knn_oracle.py, scipy k-d tree in float64)."""
import numpy as np
import pytest

from oracle import ts_knn_oracle as KO

pytestmark = pytest.mark.gpu


def _pts(n, seed, kind="uniform"):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, 3), dtype=np.float32) * np.array([10, 6, 3], np.float32)
    if kind == "clustered":  # most points in tight clusters, a few far away: long box lists, uneven radii
        c = rng.normal(size=(8, 3)).astype(np.float32) * 20
        p = c[rng.integers(0, 8, n)] + rng.normal(size=(n, 3)).astype(np.float32) * 0.05
        p[: n // 50] = rng.normal(size=(n // 50, 3)).astype(np.float32) * 100
        return p.astype(np.float32)
    raise ValueError(kind)


@pytest.mark.parametrize("n,kind", [(9, "uniform"), (1000, "uniform"), (1025, "uniform"), (5000, "clustered"),
                                    (200_000, "uniform"), (100_000, "clustered")])
def test_mean_dist3_matches_exact_search(n, kind):
    import torch
    from simple_knn import distCUDA2
    p = _pts(n, n, kind)
    got = distCUDA2(torch.from_numpy(p).cuda()).cpu().numpy().astype(np.float64)
    want = KO.mean_dist3(p)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-12)


@pytest.mark.parametrize("n,g,kind", [(9, 3, "uniform"), (3000, 3, "uniform"), (3072, 1, "uniform"), (60_000, 3, "clustered"),
                                      (150_000, 3, "uniform"), (4096, 1024, "uniform")])
def test_nearest_other_matches_exact_search(n, g, kind):
    import torch
    from simple_knn import nearestNeighbor
    p = _pts(n, 7 * n + g, kind)
    got = nearestNeighbor(torch.from_numpy(p).cuda(), g)
    assert got.dtype == torch.uint32
    got = got.view(torch.int32).cpu().numpy().astype(np.int64)
    idx, d2 = KO.nearest_other(p, g)
    assert ((got // g) != (np.arange(n) // g)).all()
    same = got == idx
    # where the index differs the distances must tie to fp32 rounding (the search is exact, ties are broken by Morton order)
    dg = ((p[got].astype(np.float64) - p.astype(np.float64)) ** 2).sum(1)
    assert np.all(same | (np.abs(dg - d2) <= 1e-6 * np.maximum(d2, 1e-12)))
    assert same.mean() > 0.999


def test_reference_demo_points():
    """The nine points of submodules/simple-knn/main.cu:53-61 (three clusters of three), results worked out by hand."""
    import torch
    from simple_knn import distCUDA2, nearestNeighbor
    p = np.array([[0, 0, .1], [.5, 0, 0], [0, 1, 0], [0, 3, .1], [.5, 3, 0], [0, 4, 0], [3, 0, .1], [3.5, 0, 0], [3, 1, 0]], np.float32)
    t = torch.from_numpy(p).cuda()
    d = distCUDA2(t).cpu().numpy()
    np.testing.assert_allclose(d, KO.mean_dist3(p), rtol=1e-6)
    nn = nearestNeighbor(t, 3).view(torch.int32).cpu().numpy()
    # nearest point of ANOTHER triple: cluster {0,1,2} <-> {3,4,5} along y, {6,7,8} reaches back to {0,1,2} along x
    idx, d2 = KO.nearest_other(p, 3)
    dg = ((p[nn].astype(np.float64) - p) ** 2).sum(1)
    np.testing.assert_allclose(dg, d2, rtol=1e-6)
    assert nn[2] == 3 and nn[3] == 2 and nn[6] == 1
    # point 0 is exactly 3 away from both point 3 (0, 3, .1) and point 6 (3, 0, .1): the reference keeps the first one met
    # in Morton order (strict `<`, simple_knn.cu:229).  With the origin-seeded box (0,0,0)-(3.5,4,.1) the leading
    # (z, y, x) bit triples are (1,1,0) for point 3 and (1,0,1) for point 6, so point 6 sorts first.
    assert nn[0] == 6


def test_small_inputs_and_errors():
    import torch
    from simple_knn import distCUDA2, nearestNeighbor
    assert distCUDA2(torch.zeros((0, 3), device="cuda")).shape == (0,)
    one = distCUDA2(torch.rand((1, 3), device="cuda"))
    assert torch.isinf(one).all()  # three missing neighbours: FLT_MAX * 3 overflows like the reference
    three = distCUDA2(torch.tensor([[0., 0, 0], [1, 0, 0], [0, 2, 0]], device="cuda")).cpu().numpy()
    np.testing.assert_allclose(three, KO.mean_dist3(np.array([[0., 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)), rtol=1e-6)
    dup = torch.tensor([[1., 1, 1]] * 5, device="cuda")  # duplicates are neighbours at distance 0
    assert (distCUDA2(dup) == 0).all()
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros((4, 2), device="cuda"))
    with pytest.raises(RuntimeError):
        nearestNeighbor(torch.zeros((4, 3), device="cuda"), 3)
    with pytest.raises(RuntimeError):
        nearestNeighbor(torch.zeros((4, 3), device="cuda"), 0)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros((4, 3)))
