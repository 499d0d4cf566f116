"""The hand-written binning primitives (csrc/binning.hip) against AMD's rocPRIM on the same arrays and against numpy:
stable LSD radix sort of (key, value) pairs (replaces cub::DeviceRadixSort::SortPairs, R2D/src/rasterizer.cu:210-218) and the
tile-count prefix sum (replaces cub::DeviceScan::InclusiveSum, :186), bit for bit."""
import ctypes as C

import numpy as np
import pytest

import helpers
import synthetic

pytestmark = pytest.mark.gpu


def _lib():
    return helpers.lab_library()  # the sort / scan hooks and the rocPRIM comparators live in tools/bin/libts2d_lab.so (csrc/ts2d_lab.h)


def _sort(keys, vals, end_bit, which):
    import torch
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int32)).cuda()
    ko, vo = torch.empty_like(k), torch.empty_like(v)
    stream = torch.cuda.current_stream().cuda_stream
    assert _lib().ts2d_test_sort_pairs(k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(), k.numel(), end_bit, which, stream) == 0
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n,end_bit,kind", [
    (1, 8, "uniform"), (63, 13, "uniform"), (64, 5, "uniform"), (1023, 32, "uniform"), (1024, 32, "uniform"), (1025, 32, "uniform"),
    (65_537, 13, "uniform"),          # 1080p tile bits, one chunk past a slab of 64 chunks
    (300_000, 32, "depth"),           # float depth keys (few distinct exponents), many exact ties
    (1_000_000, 32, "depth"),
    (4_610_735, 13, "tiles"),         # the headline scene's instance count and tile count
    (200_000, 17, "uniform"),         # 3 passes (> 65536 tiles)
    (50_000, 9, "constant"),          # every key equal: the pass must keep the input order
])
def test_radix_sort_matches_rocprim_and_numpy(n, end_bit, kind):
    rng = np.random.default_rng(n + end_bit)
    if kind == "depth":
        keys = (1000.0 + 200.0 * rng.random(n, dtype=np.float32)).astype(np.float32)
        keys[rng.random(n) < 0.05] = 0.0          # culled triangles
        keys[rng.integers(0, n, n // 10)] = keys[rng.integers(0, n, n // 10)]  # exact ties
        keys = keys.view(np.uint32)
    elif kind == "tiles":
        keys = rng.integers(0, 8160, n, dtype=np.uint32)
    elif kind == "constant":
        keys = np.full(n, 0x155 & ((1 << end_bit) - 1), np.uint32)
    else:
        keys = rng.integers(0, 1 << end_bit, n, dtype=np.uint64).astype(np.uint32)
    vals = rng.permutation(n).astype(np.uint32)
    ours = _sort(keys, vals, end_bit, 0)
    tickets = _sort(keys, vals, end_bit, 2)  # the hierarchical (ticket) passes that sorts of more than 48 slabs fall back to
    assert np.array_equal(ours[0], tickets[0]) and np.array_equal(ours[1], tickets[1])
    theirs = _sort(keys, vals, end_bit, 1)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ours[0], keys[order]) and np.array_equal(ours[1], vals[order])
    assert np.array_equal(ours[0], theirs[0]) and np.array_equal(ours[1], theirs[1])


@pytest.mark.parametrize("P,W,H", [(5, 64, 64), (1000, 300, 200), (1025, 128, 128), (300_000, 800, 800), (1_000_000, 1920, 1080),
                                   (3_000_000, 1920, 1080)])  # > 2048 scan blocks and > 48 slabs of instances: the ticket versions of scan and tile sort
def test_instance_offsets_match_rocprim_scan(P, W, H):
    """offsets = inclusive prefix sum of tiles_touched in depth order (block sums + DPP wave scans fused into the emission
    kernel) against rocPRIM's inclusive_scan of the same counts; N = the last offset."""
    import torch
    s = synthetic.scene(P, W, H, 0, seed=P)
    hf = helpers.hip_forward_backward(s, rich_info=True, backward=False)
    perm = helpers.hip_state(hf, s, "depth_perm").astype(np.int64)
    tt = helpers.hip_state(hf, s, "tiles_touched").astype(np.uint32)
    counts = torch.from_numpy(tt[perm].view(np.int32)).cuda()
    out = torch.empty_like(counts)
    assert _lib().ts2d_test_inclusive_scan_rocprim(counts.data_ptr(), out.data_ptr(), P, torch.cuda.current_stream().cuda_stream) == 0
    off = helpers.hip_state(hf, s, "point_offsets").view(np.uint32)
    assert np.array_equal(off, out.cpu().numpy().view(np.uint32))
    assert int(off[-1]) == hf["num_rendered"]
    # the emitted list is sorted by (tile, depth, id) and the ranges partition it
    keys = helpers.hip_state(hf, s, "keys")
    assert np.all(np.diff(keys) >= 0)
    ranges = helpers.hip_state(hf, s, "ranges").astype(np.int64)
    assert np.array_equal(ranges[:, 1] - ranges[:, 0], np.bincount(keys >> 32, minlength=ranges.shape[0]))


@pytest.mark.parametrize("P", [1, 63, 64, 65, 1000, 1024, 1025, 4097, 10_000, 12_287, 12_288, 12_289, 20_000])
@pytest.mark.parametrize("spread", [False, True])
def test_depth_order_of_small_scenes_is_the_stable_sort(P, spread):
    """Up to 12 288 triangles ONE launch orders the triangles (binning.hip: depth_order_small_kernel -- one workgroup, the pairs in registers, LDS
    between the passes) and leaves what the census and the block-sum launch leave at larger sizes; above, the multi-launch sort.  Either way:
    ids in (depth bits, id) order = numpy's stable argsort of the keys (culled triangles carry key 0), offsets = the running sum of the tile
    counts in that order, N = its last value.  `spread`: depths over a factor > 4, so the top key byte varies and the fourth pass runs."""
    s = synthetic.scene(P, 192, 160, 0, seed=4000 + P)
    if spread:
        rng = np.random.default_rng(P)
        cam = np.asarray(s["campos"], np.float32).reshape(1, 1, 3)  # every triangle moved along its ray from the camera: same pixels, other depth
        s["vertex"] = (cam + (s["vertex"] - cam) * rng.uniform(0.2, 3.0, size=(P, 1, 1))).astype(np.float32)
    hf = helpers.hip_forward_backward(s, rich_info=True, backward=False)
    depth = helpers.hip_state(hf, s, "depth").view(np.uint32)
    radii = hf["radii"].reshape(-1)
    keys = np.where(radii > 0, depth, 0).astype(np.uint32)
    perm = helpers.hip_state(hf, s, "depth_perm").astype(np.int64)
    assert np.array_equal(perm, np.argsort(keys, kind="stable"))
    tt = helpers.hip_state(hf, s, "tiles_touched").astype(np.int64)
    off = helpers.hip_state(hf, s, "point_offsets").view(np.uint32).astype(np.int64)
    assert np.array_equal(off, np.cumsum(tt[perm]))
    assert int(off[-1]) == hf["num_rendered"]
    if spread:
        vis = keys[keys != 0]
        assert vis.size == 0 or (vis.max() >> 24) != (vis.min() >> 24) or P < 64  # the fourth pass had something to order


@pytest.mark.parametrize("P", [1, 3, 8, 9, 40])
@pytest.mark.parametrize("variant", [2, 3])
def test_tiny_scene_in_recycled_state_buffers(P, variant):
    """The state buffers come from torch's caching allocator, i.e. with whatever the previous owner left in them: the binning
    kernels' tickets must be zeroed by the step itself even when the scene has fewer triangles than there are tickets
    (the fuzz sweep's P = 1 cases aborted when they ran after a larger case)."""
    import torch
    s = synthetic.scene(P, 129, 5, 1, seed=77 + P, edge_px=20.0)
    of = helpers.oracle_forward(s, True, False, variant=variant)
    for _ in range(3):
        junk = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for n in (64, 1024, 4096, 65536, 1 << 20)]
        del junk
        hf = helpers.hip_forward_backward(s, True, False, variant=variant)
        assert hf["num_rendered"] == of["num_rendered"]
        assert helpers.rel_l2(hf["out_feature"], of["out_feature"]) < 1e-4


def test_last_arrival_handoffs_under_uneven_load():
    """Every block-to-block hand-off of the binning kernels (per-slab / per-pass prefixes of the radix sort, the scan's block sums: write-through
    stores, a drained ticket, sc1 loads in the last-arriving block -- a gfx950 hardware contract, binning.hip) repeated under UNEVEN load:
    a second stream keeps some CUs busy with streaming copies and matrix products while sorts of several sizes run back to back on
    the first, so that producers and the elected consumer meet on busy and idle CUs, same and different XCDs, warm and cold L1s.
    Every result is compared bit for bit against numpy's stable sort; 60 sorts x up to 6 passes x 2 hand-off levels."""
    import torch
    rng = np.random.default_rng(2026)
    side = torch.cuda.Stream()
    big = torch.empty((64 << 20,), device="cuda", dtype=torch.float32)
    a = torch.randn((2048, 2048), device="cuda")
    stop = torch.cuda.Event()
    sizes = [65_537, 262_145, 1_000_003, 300_007, 4_100_001, 70_001]
    bad = 0
    for it in range(60):
        n = sizes[it % len(sizes)]
        end_bit = [13, 32, 21, 32, 13, 8][it % 6]
        keys = rng.integers(0, 2 ** min(end_bit, 32), n, dtype=np.uint64).astype(np.uint32)
        if it % 3 == 0:
            keys = (keys >> 7) << 7  # long runs of equal digits in the low passes
        vals = np.arange(n, dtype=np.uint32)
        with torch.cuda.stream(side):  # uneven neighbour load: a few large copies and GEMMs of varying length
            for _ in range(1 + it % 4):
                big[: (16 << 20) * (1 + it % 3)].mul_(1.0001)
                a = (a @ a).clamp_(-1, 1)
        ko, vo = _sort(keys, vals, end_bit, 2 if it % 2 == 0 else 0)  # the ticket passes and the ticket-free ones (atomics into slab totals)
        order = np.argsort(keys, kind="stable")
        if not (np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])):
            bad += 1
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 60 sorts differ from the stable reference under load"


@pytest.mark.parametrize("n", [1, 2, 17, 4096, 4097, 640_000, 1 << 20, (1 << 20) + 1, 3_000_001])
def test_radix_select_quantile_equals_torch_quantile(n):
    """select.hip: torch.quantile(G, q) of non-negative floats by a most-significant-digit radix select (four histogram passes + a neighbour pass)
    against torch.quantile itself (linear interpolation), on uniform values, values spread over many octaves, a few distinct values with heavy
    duplicates, and mostly zeros; q at the ends and inside; sizes around the kernels' block and chunk boundaries.  (Round 6 also built the five
    launches as ONE, the workgroups handing the digit totals to each other through a counter: 37 -> 46 us at 640 k keys, 96 with the totals spread
    over eight copies -- a hand-off between workgroups costs more than a launch on this chip.  Not adopted.)"""
    import torch
    L = _lib()
    g = torch.Generator(device="cuda").manual_seed(n)
    base = torch.rand(n, device="cuda", generator=g)
    data = {"uniform": base, "octaves": torch.exp2(40.0 * base - 30.0), "duplicates": torch.round(base * 7.0) / 7.0,
            "zeros": torch.where(base < 0.95, torch.zeros_like(base), base)}
    scratch = torch.empty(L.ts2d_test_quantile_scratch_bytes(), device="cuda", dtype=torch.uint8)
    out = torch.empty(1, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for name, x in data.items():
        x = x.float().contiguous()
        for q in (0.0, 0.1, 0.5, 0.9, 0.999, 1.0):
            want = float(torch.quantile(x, q))
            assert L.ts2d_test_quantile(x.data_ptr(), n, q, scratch.data_ptr(), out.data_ptr(), stream) == 0
            got = float(out)
            assert got == want or abs(got - want) <= 2e-7 * abs(want), (name, q, got, want)


@pytest.mark.parametrize("seed", range(12))
def test_depth_order_of_random_scenes_is_the_stable_sort_of_the_depth_keys(seed):
    """The product's three depth-order forms by size -- one launch up to 9 216 triangles, sampled splitters + per-bucket sorts to 500 000, LSD passes
    beyond -- on random sizes around their switch-overs and random depth layouts (plain, quantised to a few values, a far cluster, depths over several
    octaves, a share of culled triangles): the permutation must be numpy's STABLE argsort of the depth keys (bit patterns of the view-space depth, 0
    for culled triangles: rasterizer.cu:211's SortPairs is stable), and the instance offsets its inclusive scan of the tile counts in that order."""
    rng = np.random.default_rng(1000 + seed)
    P = int([rng.integers(2, 9216), rng.integers(9217, 40000), rng.integers(40000, 300000), rng.integers(300000, 500001),
             rng.integers(500001, 700000), 9216, 9217, 500000, 500001][seed % 9])
    s = synthetic.scene(P, 256, 160, 0, seed=int(rng.integers(1 << 30)))
    v = s["vertex"]
    kind = ["plain", "quantised", "far", "octaves", "culled", "one_depth"][seed % 6]
    zc = v[:, :, 2].mean(axis=1, keepdims=True)
    if kind == "quantised":
        v[:, :, 2] = np.round(zc / 15.0) * 15.0
    elif kind == "far":
        v[: max(1, P // 40), :, 2] -= 20000.0
    elif kind == "octaves":
        scale = np.exp2(rng.uniform(-3.0, 0.0, size=(P, 1))).astype(np.float32)
        v[:, :, 2] = s["campos"][2] + (v[:, :, 2] - s["campos"][2]) * scale
    elif kind == "culled":
        v[rng.random(P) < 0.3, :, 2] += 5000.0
    elif kind == "one_depth":
        v[: P // 2, :, 2] = zc[: P // 2].mean()
    hf = helpers.hip_forward_backward(s, True, backward=False)
    keys = helpers.hip_state(hf, s, "depth").view(np.uint32)
    perm = helpers.hip_state(hf, s, "depth_perm")
    want = np.argsort(keys, kind="stable")
    assert np.array_equal(perm.astype(np.int64), want), (P, kind)
    tiles = helpers.hip_state(hf, s, "tiles_touched").astype(np.int64)
    assert np.array_equal(helpers.hip_state(hf, s, "point_offsets").astype(np.int64), np.cumsum(tiles[want])), (P, kind)
    assert int(hf["num_rendered"]) == int(tiles.sum())
