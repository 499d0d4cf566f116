// resample.hip -- the down-sampler of render_up_scale (include/ts_loss.h: tsl_downsample_forward / _backward).
//
// The reference renders at s x the camera's resolution and resizes render, depth and normal back with
// torch.nn.functional.interpolate(..., size=(h, w), mode="bilinear") (src/diff_recon/models/VanillaTS_model.py:625-630, 649-656;
// render_up_scale = 2 in the NerfSynthetic *_mesh configuration).  With an INTEGER factor s every output pixel reads at most 2 x 2 input
// pixels -- PyTorch's source index (d + 0.5) s - 0.5 (align_corners = False, no antialias), s = 2: the mean of a 2 x 2 block -- and every input
// pixel feeds at most ONE output pixel, so the backward is a gather too: no atomics (torch's upsample_bilinear2d_backward adds with atomics),
// run-to-run identical, both directions one HBM-bound pass (forward s^2 + 1 floats per output pixel, backward 1 + s^2).
// Weights are formed like ATen's area_pixel_compute_source_index / compute_source_index_and_lambda in float32 (scale = (float)in / out).
#include "../../include/ts_loss.h"
#include "ts2d_common.h"

namespace
{
// One launch serves up to TS_RESAMPLE_PLANES image planes that need not be contiguous with each other (blockIdx.z = plane): the render (3), depth (1)
// and normal (3) images of a step are three tensors, and at 800 x 800 a launch is ~9 us of latency for ~3 of work (round 6: six launches -> two).
constexpr int PLANES = TS_RESAMPLE_PLANES;
struct PlanePtrs
{
    const float *src[PLANES];
    float *dst[PLANES];
};
struct Tap { int i0, i1; float w0, w1; };
__device__ __forceinline__ Tap tap_of(int d, float scale, int in_size)
{
    // aten/src/ATen/native/UpSample.h: real = scale * (d + 0.5) - 0.5, clamped at 0; i0 = (int)real; lambda1 = real - i0; i1 = i0 + (i0 < in - 1)
    float real = scale * ((float)d + 0.5f) - 0.5f;
    real = real < 0.0f ? 0.0f : real;
    Tap t;
    t.i0 = (int)real;
    if (t.i0 > in_size - 1) t.i0 = in_size - 1;
    t.i1 = t.i0 + (t.i0 < in_size - 1 ? 1 : 0);
    t.w1 = real - (float)t.i0;
    t.w0 = 1.0f - t.w1;
    return t;
}

__global__ void __launch_bounds__(256) downsample_fwd_kernel(PlanePtrs pl, int H, int W, int h, int w, float sy, float sx)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const Tap ty = tap_of(y, sy, H), tx = tap_of(x, sx, W);
    const float *p = pl.src[blockIdx.z];
    const float a = p[(size_t)ty.i0 * W + tx.i0], b = p[(size_t)ty.i0 * W + tx.i1], cc = p[(size_t)ty.i1 * W + tx.i0], d = p[(size_t)ty.i1 * W + tx.i1];
    // upsample_bilinear2d_out_frame: h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (w0lambda * c + w1lambda * d)
    pl.dst[blockIdx.z][(size_t)y * w + x] = ty.w0 * (tx.w0 * a + tx.w1 * b) + ty.w1 * (tx.w0 * cc + tx.w1 * d);
}

// dL/d in[Y][X] = sum over the (at most one, for an integer factor >= 2) output pixels whose taps include (Y, X)
__global__ void __launch_bounds__(256) downsample_bwd_kernel(PlanePtrs pl, int H, int W, int h, int w, float sy, float sx, int fy, int fx)
{
    const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= W || Y >= H) return;
    // candidates: the output pixels around Y / fy (an output pixel's taps lie inside its own block of fy input rows for even factors, and are
    // the block's centre row for odd ones; the neighbours are checked as well so that the edge clamps are covered)
    float wy[3], wx[3];
    int ys[3], xs[3];
#pragma unroll
    for (int k = 0; k < 3; k++)
    {
        const int y = Y / fy - 1 + k, x = X / fx - 1 + k;
        ys[k] = y; xs[k] = x;
        wy[k] = wx[k] = 0.0f;
        if (y >= 0 && y < h) { const Tap t = tap_of(y, sy, H); wy[k] = (t.i0 == Y ? t.w0 : 0.0f) + (t.i1 == Y ? t.w1 : 0.0f); }
        if (x >= 0 && x < w) { const Tap t = tap_of(x, sx, W); wx[k] = (t.i0 == X ? t.w0 : 0.0f) + (t.i1 == X ? t.w1 : 0.0f); }
    }
    const float *g = pl.src[blockIdx.z];
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            if (wy[j] != 0.0f && wx[i] != 0.0f) acc += wy[j] * wx[i] * g[(size_t)ys[j] * w + xs[i]];
    pl.dst[blockIdx.z][(size_t)Y * W + X] = acc;
}
// Factor 2 in both directions (render_up_scale = 2, the configuration that ships): both taps have weight 0.5 in float32 exactly as the general
// kernels compute them, so the results are the same bits; one thread per OUTPUT pixel, the 2 x 2 input block as two float2.
__global__ void __launch_bounds__(256) downsample2_fwd_kernel(PlanePtrs pl, int h, int w)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int W = 2 * w;
    const float *p = pl.src[blockIdx.z] + (size_t)(2 * y) * W + 2 * x;
    const float2 r0 = *(const float2 *)p, r1 = *(const float2 *)(p + W);
    pl.dst[blockIdx.z][(size_t)y * w + x] = 0.5f * (0.5f * r0.x + 0.5f * r0.y) + 0.5f * (0.5f * r1.x + 0.5f * r1.y);
}
__global__ void __launch_bounds__(256) downsample2_bwd_kernel(PlanePtrs pl, int h, int w)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int W = 2 * w;
    const float g = 0.5f * 0.5f * pl.src[blockIdx.z][(size_t)y * w + x]; // wy * wx * g, the general kernel's product
    float *p = pl.dst[blockIdx.z] + (size_t)(2 * y) * W + 2 * x;
    *(float2 *)p = make_float2(g, g);
    *(float2 *)(p + W) = make_float2(g, g);
}
} // namespace

// `src` / `dst`: n plane pointers each (device memory; any n >= 0: groups of TS_RESAMPLE_PLANES per launch)
hipError_t ts_downsample_forward_planes(int n, const float *const *src, int H, int W, int h, int w, float *const *dst, hipStream_t s)
{
    for (int c0 = 0; c0 < n; c0 += PLANES)
    {
        const int m = n - c0 < PLANES ? n - c0 : PLANES;
        PlanePtrs pl{};
        bool fast = H == 2 * h && W == 2 * w;
        for (int k = 0; k < m; k++) { pl.src[k] = src[c0 + k]; pl.dst[k] = dst[c0 + k]; fast = fast && ((size_t)pl.src[k] & 7) == 0; }
        const dim3 grid((w + 63) / 64, (h + 3) / 4, m);
        if (fast) hipLaunchKernelGGL(downsample2_fwd_kernel, grid, dim3(256), 0, s, pl, h, w);
        else hipLaunchKernelGGL(downsample_fwd_kernel, grid, dim3(256), 0, s, pl, H, W, h, w, (float)H / (float)h, (float)W / (float)w);
    }
    return hipGetLastError();
}
// `gout`: n planes of h x w, `gin`: n planes of H x W
hipError_t ts_downsample_backward_planes(int n, const float *const *gout, int H, int W, int h, int w, float *const *gin, hipStream_t s)
{
    for (int c0 = 0; c0 < n; c0 += PLANES)
    {
        const int m = n - c0 < PLANES ? n - c0 : PLANES;
        PlanePtrs pl{};
        bool fast = H == 2 * h && W == 2 * w;
        for (int k = 0; k < m; k++) { pl.src[k] = gout[c0 + k]; pl.dst[k] = gin[c0 + k]; fast = fast && ((size_t)pl.dst[k] & 7) == 0; }
        if (fast) hipLaunchKernelGGL(downsample2_bwd_kernel, dim3((w + 63) / 64, (h + 3) / 4, m), dim3(256), 0, s, pl, h, w);
        else
            hipLaunchKernelGGL(downsample_bwd_kernel, dim3((W + 63) / 64, (H + 3) / 4, m), dim3(256), 0, s, pl, H, W, h, w, (float)H / (float)h, (float)W / (float)w,
                               H / h, W / w);
    }
    return hipGetLastError();
}

hipError_t ts_downsample_forward(const float *in, int C, int H, int W, int h, int w, float *out, hipStream_t s)
{
    for (int c0 = 0; c0 < C; c0 += PLANES) // the planes of ONE tensor
    {
        const int m = C - c0 < PLANES ? C - c0 : PLANES;
        const float *src[PLANES];
        float *dst[PLANES];
        for (int k = 0; k < m; k++) { src[k] = in + (size_t)(c0 + k) * H * W; dst[k] = out + (size_t)(c0 + k) * h * w; }
        if (hipError_t e = ts_downsample_forward_planes(m, src, H, W, h, w, dst, s)) return e;
    }
    return hipSuccess;
}

hipError_t ts_downsample_backward(const float *gout, int C, int H, int W, int h, int w, float *gin, hipStream_t s)
{
    for (int c0 = 0; c0 < C; c0 += PLANES)
    {
        const int m = C - c0 < PLANES ? C - c0 : PLANES;
        const float *src[PLANES];
        float *dst[PLANES];
        for (int k = 0; k < m; k++) { src[k] = gout + (size_t)(c0 + k) * h * w; dst[k] = gin + (size_t)(c0 + k) * H * W; }
        if (hipError_t e = ts_downsample_backward_planes(m, src, H, W, h, w, dst, s)) return e;
    }
    return hipSuccess;
}
