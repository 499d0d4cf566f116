"""The 16.48 fixed-point conversion behind the forward's contribution statistics (csrc/ts2d_group.h: to_fixed48), restated with
numpy float32 / Python integers: the three device operations (a multiplication by 2^48, a truncating conversion of y * 2^-32, a fused
multiply-add that forms y - hi * 2^32) are exact for every float32 the kernels can produce, so the LDS sums are exact sums of the
contributions truncated to 2^-48 -- closer to the reference's result than any fp32 summation order is to another."""
import numpy as np


def to_fixed48(x32: np.ndarray) -> np.ndarray:
    x = x32.astype(np.float32)
    y = (x * np.float32(2.0 ** 48)).astype(np.float32)                  # exact: a power-of-two scaling inside the float32 range
    hi = np.floor((y * np.float32(2.0 ** -32)).astype(np.float32)).astype(np.uint64)   # v_cvt_u32_f32 truncates
    rem = y.astype(np.float64) - hi.astype(np.float64) * 2.0 ** 32     # the fma's infinitely precise result ...
    assert np.array_equal(rem.astype(np.float32).astype(np.float64), rem)  # ... is a float32: nothing is rounded
    lo = np.floor(rem).astype(np.uint64)
    return (hi << np.uint64(32)) | lo


def test_conversion_is_the_floor_of_x_times_2_to_the_48():
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.random(200_000, dtype=np.float32) * 16.0,                      # a (group, entry) sum is at most 16
        np.exp(rng.uniform(np.log(1e-4 / 255), np.log(16.0), 200_000)).astype(np.float32),  # down to the smallest contribution
        np.array([1e-4 / 255, 2.0 ** -25, 1.0, 15.999999, 16.0, 0.99 * 2.0 ** -13], np.float32),
    ])
    got = to_fixed48(x)
    want = np.array([int(float(v) * 2 ** 48) for v in x.astype(np.float64)], dtype=np.uint64)  # Python ints: exact
    assert np.array_equal(got, want)
    # values with at most 24 significant bits above 2^-48 are represented exactly (everything >= 2^-25)
    big = x >= np.float32(2.0 ** -25)
    assert np.array_equal(got[big].astype(np.float64) * 2.0 ** -48, x[big].astype(np.float64))
    # the smallest contribution the blend can produce (alpha = 1/255 at T = 1e-4) keeps 27 significant bits
    assert abs(float(to_fixed48(np.array([1e-4 / 255], np.float32))[0]) * 2.0 ** -48 / np.float32(1e-4 / 255) - 1.0) < 2.0 ** -26


def test_sums_do_not_depend_on_the_order_and_fit():
    rng = np.random.default_rng(1)
    c = (rng.random(4096, dtype=np.float32) * 0.06).astype(np.float32)      # a tile: 256 pixels x 16 entries' worth of contributions
    a = int(to_fixed48(c).sum(dtype=np.uint64))
    b = int(to_fixed48(c[rng.permutation(len(c))]).sum(dtype=np.uint64))
    assert a == b                                                          # integer addition: any order, any grouping
    exact = float(np.sum(c.astype(np.float64)))
    assert abs(a * 2.0 ** -48 - exact) <= len(c) * 2.0 ** -48               # truncation of at most one unit per addend
    assert 256 * 2 ** 48 < 2 ** 63                                          # a tile's sum (<= 256) is far from the 64-bit range
