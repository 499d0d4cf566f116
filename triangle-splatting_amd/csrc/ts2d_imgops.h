// ts2d_imgops.h -- image operators shared by the loss kernels (depth_normal.hip, aux_losses.hip): PyTorch's bilinear resampling taps
// (F.interpolate, align_corners = False), the zero-padded Scharr pair of the reference's ScharrFilter (trainer_utils.py:151-178), and the
// torch.quantile threshold of an ascending array.  Moved out of depth_normal.hip in round 5, unchanged.
#pragma once
#include "ts2d_common.h"

namespace
{
// PyTorch's bilinear source index (align_corners = False; aten/native/UpSample.h area_pixel_compute_source_index + guard_index_and_lambda)
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap tap_of(int p, float r, int S)
{
    float src = r * ((float)p + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    Tap t;
    t.i0 = min((int)src, S - 1);
    t.i1 = t.i0 + (t.i0 < S - 1 ? 1 : 0);
    t.l1 = fminf(fmaxf(src - (float)t.i0, 0.0f), 1.0f);
    t.l0 = 1.0f - t.l1;
    return t;
}
// destination indices whose taps can touch source index i (a superset; the caller tests each)
__device__ __forceinline__ void dst_range(int i, float r, int D, int &lo, int &hi)
{
    lo = max(0, (int)floorf(((float)i - 0.5f) / r - 0.5f) - 1);
    hi = min(D - 1, (int)ceilf(((float)i + 1.5f) / r - 0.5f) + 1);
}

__device__ __forceinline__ float at0(const float *a, int i, int j, int h, int w) { return (i >= 0 && i < h && j >= 0 && j < w) ? a[(size_t)i * w + j] : 0.0f; }
__device__ __forceinline__ void scharr(const float *d, int i, int j, int h, int w, float &gx, float &gy)
{
    const float a = at0(d, i - 1, j - 1, h, w), b = at0(d, i - 1, j, h, w), c = at0(d, i - 1, j + 1, h, w);
    const float e = at0(d, i, j - 1, h, w), f = at0(d, i, j + 1, h, w);
    const float g = at0(d, i + 1, j - 1, h, w), hh = at0(d, i + 1, j, h, w), k = at0(d, i + 1, j + 1, h, w);
    gx = (-3.0f * a + 3.0f * c - 10.0f * e + 10.0f * f - 3.0f * g + 3.0f * k) * (1.0f / 32.0f);
    gy = (-3.0f * a - 10.0f * b - 3.0f * c + 3.0f * g + 10.0f * hh + 3.0f * k) * (1.0f / 32.0f);
}

__device__ __forceinline__ float bilerp(const float *a, int w, const Tap &ty, const Tap &tx)
{
    const float *r0 = a + (size_t)ty.i0 * w, *r1 = a + (size_t)ty.i1 * w;
    return ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
}
// torch.quantile(G, q), interpolation = "linear": rank = q (n - 1) in float32 like torch, lerp between the two neighbours
__global__ void quantile_threshold_kernel(const uint32_t *__restrict__ sorted, int n, float q, float *__restrict__ thr)
{
    const float rank = q * (float)(n - 1);
    const int lo = (int)floorf(rank), hi = min((int)ceilf(rank), n - 1);
    const float a = __uint_as_float(sorted[lo]), b = __uint_as_float(sorted[hi]), wgt = rank - (float)lo;
    *thr = (wgt < 0.5f) ? a + wgt * (b - a) : b - (b - a) * (1.0f - wgt); // at::lerp
}

} // namespace
