// aux_losses.hip -- the two auxiliary image losses of the reference's trainers, forward and backward (include/ts_loss.h, round 5).
//
//   DoGLoss(freq = 90, scale_factor = 0.5)            src/diff_recon/trainers/trainer_utils.py:105-148   (w_dog, VanillaTS_trainer.py:26-27, 82, 111)
//     mask = [normalised DoG of the down-sampled grey target >= 0.5]  (no gradient),  loss = L1(img * mask, gt * mask)
//   SmoothnessLoss(quantile = 0.3, scale_factor = 0.5) trainer_utils.py:181-201                           (w_smoothness, :28-29, 83, 111)
//     mask = [up-sampled |Scharr| of the down-sampled target < its quantile]  (no gradient),  loss = mean(|Scharr(img)|_2 * mask)
// Both weights are 0 in every configuration the reference ships; the classes exist, so their counterparts do (the rest of
// trainer_utils.py's loss file).  Reference = eager torch (interpolate, depthwise conv2d, min / max or quantile, elementwise); here: the masks in
// five / six small launches (bilinear taps as PyTorch forms them, ts2d_imgops.h; the quantile from the library's own radix sort), each loss in two
// (deterministic two-stage sums in double), each gradient in one or two gather-form kernels -- no atomics, run-to-run identical.  HBM-bound.
#include "../../include/ts_loss.h"
#include "ts2d_common.h"
#include "ts2d_imgops.h"
#include <algorithm>

namespace
{
constexpr int SUM_BLOCKS = 1024;
constexpr int MAXC = 8; // channels of an image handed to these losses (api.hip checks)
struct ADims { int C, H, W, h, w; float r_down, r_up_y, r_up_x; };
struct Gauss { int k1, k2; float w1[33], w2[33]; }; // 1-D factors of the two normalised Gaussian kernels (GaussianSmoothing2D, trainer_utils.py:9-44)

ADims make_adims(int C, int H, int W, double scale)
{
    ADims m;
    m.C = C; m.H = H; m.W = W;
    const bool same = !(scale > 0.0) || scale == 1.0;
    m.h = same ? H : (int)floor((double)H * scale); // F.interpolate(scale_factor = s): floor(H * s) rows, coordinates mapped with 1 / s
    m.w = same ? W : (int)floor((double)W * scale);
    m.r_down = same ? 1.0f : (float)(1.0 / scale);
    m.r_up_y = (float)m.h / (float)H;               // F.interpolate(size = (H, W)): the size ratio maps the coordinates
    m.r_up_x = (float)m.w / (float)W;
    return m;
}

// The Scharr pair as differences of opposite taps: on a locally constant image every difference is an exact 0, so the norm is an exact 0 and its
// gradient 0 (torch's norm backward).  Summed tap by tap -- like ts2d_imgops.h's scharr, or the reference's float32 convolution -- the six products
// leave ~1 ulp of rounding there, and the gradient of the norm is then a unit vector in the direction of that noise.
__device__ __forceinline__ void scharr_sym(const float *d, int i, int j, int h, int w, float &gx, float &gy)
{
    const float a = at0(d, i - 1, j - 1, h, w), b = at0(d, i - 1, j, h, w), c = at0(d, i - 1, j + 1, h, w);
    const float e = at0(d, i, j - 1, h, w), f = at0(d, i, j + 1, h, w);
    const float g = at0(d, i + 1, j - 1, h, w), hh = at0(d, i + 1, j, h, w), k = at0(d, i + 1, j + 1, h, w);
    gx = (3.0f * (c - a) + 10.0f * (f - e) + 3.0f * (k - g)) * (1.0f / 32.0f);
    gy = (3.0f * (g - a) + 10.0f * (hh - b) + 3.0f * (k - c)) * (1.0f / 32.0f);
}

// ---- masks ---------------------------------------------------------------------------------------------------------------------------
// low-resolution planes of the target: GREY = true: one plane, the channel mean (trainer_utils.py:133) taken BEFORE the resampling like the
// reference; GREY = false: C planes
template <bool GREY>
__global__ void __launch_bounds__(256) aux_downsample_kernel(ADims m, const float *__restrict__ gt, float *__restrict__ low)
{
    const int k = blockIdx.x * 256 + threadIdx.x, hw = m.h * m.w;
    if (k >= hw) return;
    const int i = k / m.w, j = k - i * m.w;
    const size_t HW = (size_t)m.H * m.W;
    const bool same = m.h == m.H && m.w == m.W;
    const Tap ty = same ? Tap{i, i, 1.0f, 0.0f} : tap_of(i, m.r_down, m.H), tx = same ? Tap{j, j, 1.0f, 0.0f} : tap_of(j, m.r_down, m.W);
    if (GREY)
    {
        float g[4];
        const int ys[2] = {ty.i0, ty.i1}, xs[2] = {tx.i0, tx.i1};
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
            {
                float s = 0.0f;
                for (int c = 0; c < m.C; c++) s += gt[c * HW + (size_t)ys[a] * m.W + xs[b]];
                g[2 * a + b] = s / (float)m.C;
            }
        low[k] = ty.l0 * (tx.l0 * g[0] + tx.l1 * g[1]) + ty.l1 * (tx.l0 * g[2] + tx.l1 * g[3]);
    }
    else
        for (int c = 0; c < m.C; c++) low[(size_t)c * hw + k] = bilerp(gt + c * HW, m.W, ty, tx);
}

// difference of the two zero-padded Gaussian blurs (DoGFilter, trainer_utils.py:105-121)
__global__ void __launch_bounds__(256) aux_dog_kernel(ADims m, Gauss gs, const float *__restrict__ grey, float *__restrict__ dog)
{
    const int k = blockIdx.x * 256 + threadIdx.x, hw = m.h * m.w;
    if (k >= hw) return;
    const int i = k / m.w, j = k - i * m.w;
    float b1 = 0.0f, b2 = 0.0f;
    const int p1 = (gs.k1 - 1) / 2, p2 = (gs.k2 - 1) / 2;
    for (int a = 0; a < gs.k2; a++)
    {
        const int y = i + a - p2;
        if (y < 0 || y >= m.h) continue;
        const int a1 = a - p2 + p1; // the same row in the smaller kernel's coordinates
        float r1 = 0.0f, r2 = 0.0f;
        for (int b = 0; b < gs.k2; b++)
        {
            const int x = j + b - p2;
            if (x < 0 || x >= m.w) continue;
            const float v = grey[(size_t)y * m.w + x];
            r2 += gs.w2[b] * v;
            const int bb = b - p2 + p1;
            if (a1 >= 0 && a1 < gs.k1 && bb >= 0 && bb < gs.k1) r1 += gs.w1[bb] * v;
        }
        b2 += gs.w2[a] * r2;
        if (a1 >= 0 && a1 < gs.k1) b1 += gs.w1[a1] * r1;
    }
    dog[k] = b1 - b2;
}

// full-resolution plane U = bilinear(low) (+ the sort key of a non-negative U, + per-block minimum / maximum)
template <bool KEYS, bool MINMAX>
__global__ void __launch_bounds__(256) aux_upsample_kernel(ADims m, const float *__restrict__ low, float *__restrict__ U, uint32_t *__restrict__ key,
                                                            float *__restrict__ pmin, float *__restrict__ pmax)
{
    __shared__ float rmin[4], rmax[4];
    const int HW = m.H * m.W;
    float mn = 3.4e38f, mx = -3.4e38f;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < HW; k += gridDim.x * 256)
    {
        const int y = k / m.W, x = k - y * m.W;
        const float u = (m.h == m.H && m.w == m.W) ? low[k] : bilerp(low, m.w, tap_of(y, m.r_up_y, m.h), tap_of(x, m.r_up_x, m.w));
        U[k] = u;
        if (KEYS) key[k] = __float_as_uint(u);
        mn = fminf(mn, u); mx = fmaxf(mx, u);
    }
    if (MINMAX)
    {
        for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
        if ((threadIdx.x & 63) == 0) { rmin[threadIdx.x >> 6] = mn; rmax[threadIdx.x >> 6] = mx; }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            pmin[blockIdx.x] = fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
            pmax[blockIdx.x] = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
        }
    }
}
__global__ void __launch_bounds__(64) aux_minmax_finish_kernel(int nblocks, const float *__restrict__ pmin, const float *__restrict__ pmax, float *__restrict__ mm)
{
    float mn = 3.4e38f, mx = -3.4e38f; // one wave instead of one thread walking the partials (round 6, see depth_normal.hip: dn_finish_kernel)
    for (int b = threadIdx.x; b < nblocks; b += 64) { mn = fminf(mn, pmin[b]); mx = fmaxf(mx, pmax[b]); }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (threadIdx.x == 0) { mm[0] = mn; mm[1] = mx; }
}
// DoG: normalised = (U - min) / (max - min), inverted for freq >= 50, mask = normalised >= 0.5 (trainer_utils.py:138-143)
__global__ void __launch_bounds__(256) aux_dog_mask_kernel(int HW, const float *__restrict__ U, const float *__restrict__ mm, int invert, float *__restrict__ mask)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= HW) return;
    float n = (U[k] - mm[0]) / (mm[1] - mm[0]);
    if (invert) n = 1.0f - n;
    mask[k] = n >= 0.5f ? 1.0f : 0.0f;
}
// smoothness: Scharr norm over the 2 C gradient planes at low resolution (ScharrFilter(ret_norm = True), trainer_utils.py:151-178)
__global__ void __launch_bounds__(256) aux_scharr_norm_low_kernel(ADims m, const float *__restrict__ low, float *__restrict__ gnorm)
{
    const int k = blockIdx.x * 256 + threadIdx.x, hw = m.h * m.w;
    if (k >= hw) return;
    const int i = k / m.w, j = k - i * m.w;
    float s = 0.0f;
    for (int c = 0; c < m.C; c++)
    {
        float gx, gy;
        scharr(low + (size_t)c * hw, i, j, m.h, m.w, gx, gy);
        s += gx * gx + gy * gy;
    }
    gnorm[k] = sqrtf(s);
}
__global__ void __launch_bounds__(256) aux_below_mask_kernel(int HW, const float *__restrict__ U, const float *__restrict__ thr, float *__restrict__ mask)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < HW) mask[k] = U[k] < *thr ? 1.0f : 0.0f; // trainer_utils.py:193
}

// ---- losses --------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum_to(double s, double *partial)
{
    __shared__ double red[4];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(64) aux_finish_kernel(int nblocks, double count, const double *__restrict__ partial, float *__restrict__ out)
{
    double s = 0.0; // fixed order per lane, then a butterfly: deterministic
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[0] = (float)(s / count);
}
// L1(img * mask, gt * mask) = mean over C H W of |img m - gt m| (trainer_utils.py:147-148, 323-324); the mask is one plane
__global__ void __launch_bounds__(256) aux_masked_l1_sum_kernel(int C, int HW, const float *__restrict__ img, const float *__restrict__ gt,
                                                                 const float *__restrict__ mask, double *__restrict__ partial)
{
    double s = 0.0;
    const size_t n = (size_t)C * HW;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
    {
        const float m = mask[k % HW];
        s += (double)fabsf(img[k] * m - gt[k] * m);
    }
    block_sum_to(s, partial);
}
__global__ void __launch_bounds__(256) aux_masked_l1_bwd_kernel(int C, int HW, const float *__restrict__ img, const float *__restrict__ gt,
                                                                 const float *__restrict__ mask, const float *__restrict__ grad_out, float *__restrict__ dimg)
{
    const size_t n = (size_t)C * HW, k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const float m = mask[k % HW], d = img[k] * m - gt[k] * m, go = grad_out ? *grad_out : 1.0f;
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); // torch's abs backward: sign(0) = 0
    dimg[k] = go * sg * m / (float)n;
}
// mean over H W of |Scharr(img)|_2 * mask (trainer_utils.py:196-200): norm over the 2 C gradient planes
__global__ void __launch_bounds__(256) aux_smooth_sum_kernel(int C, int H, int W, const float *__restrict__ img, const float *__restrict__ mask,
                                                              double *__restrict__ partial)
{
    double s = 0.0;
    const int HW = H * W;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < HW; k += gridDim.x * 256)
    {
        const float m = mask[k];
        if (m == 0.0f) continue;
        const int i = k / W, j = k - i * W;
        float q = 0.0f;
        for (int c = 0; c < C; c++)
        {
            float gx, gy;
            scharr_sym(img + (size_t)c * HW, i, j, H, W, gx, gy);
            q += gx * gx + gy * gy;
        }
        s += (double)(sqrtf(q) * m);
    }
    block_sum_to(s, partial);
}
// backward, stage 1: per pixel q and channel c the adjoints of the two gradient planes, a = w gx / |.|, b = w gy / |.|, w = mask g / (H W)
// (torch's norm backward: 0 where the norm is 0)
__global__ void __launch_bounds__(256) aux_smooth_adj_kernel(int C, int H, int W, const float *__restrict__ img, const float *__restrict__ mask,
                                                              const float *__restrict__ grad_out, float *__restrict__ adj)
{
    const int HW = H * W, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= HW) return;
    const int i = k / W, j = k - i * W;
    const float m = mask[k], go = grad_out ? *grad_out : 1.0f;
    float gx[MAXC], gy[MAXC];
    float q = 0.0f;
    for (int c = 0; c < C; c++)
    {
        scharr_sym(img + (size_t)c * HW, i, j, H, W, gx[c], gy[c]);
        q += gx[c] * gx[c] + gy[c] * gy[c];
    }
    const float nrm = sqrtf(q), w = (m != 0.0f && nrm > 0.0f) ? go * m / ((float)HW * nrm) : 0.0f;
    for (int c = 0; c < C; c++)
    {
        adj[(size_t)(2 * c) * HW + k] = w * gx[c];
        adj[(size_t)(2 * c + 1) * HW + k] = w * gy[c];
    }
}
// stage 2: the Scharr adjoint as a gather (zero padding is self-adjoint), as depth_normal.hip's dn_bwd_scharr_kernel
__global__ void __launch_bounds__(256) aux_smooth_bwd_kernel(int C, int H, int W, const float *__restrict__ adj, float *__restrict__ dimg)
{
    const int HW = H * W;
    const size_t n = (size_t)C * HW, kk = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= n) return;
    const int c = (int)(kk / HW), k = (int)(kk - (size_t)c * HW), r = k / W, col = k - r * W;
    const float *ax = adj + (size_t)(2 * c) * HW, *ay = adj + (size_t)(2 * c + 1) * HW;
    float s = 0.0f;
#pragma unroll
    for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++)
        {
            const int a = 1 - dr, b = 1 - dc; // kernel element that output (r + dr, col + dc) applied to input (r, col)
            const float kx = (b == 1 ? 0.0f : (b == 0 ? -1.0f : 1.0f)) * (a == 1 ? 10.0f : 3.0f);
            const float ky = (a == 1 ? 0.0f : (a == 0 ? -1.0f : 1.0f)) * (b == 1 ? 10.0f : 3.0f);
            s += (kx * at0(ax, r + dr, col + dc, H, W) + ky * at0(ay, r + dr, col + dc, H, W)) * (1.0f / 32.0f);
        }
    dimg[kk] = s;
}

struct ACarve
{
    float *low, *low1, *U, *mm, *pmin, *pmax, *thr, *adj;
    uint32_t *k[2], *v[2];
    double *partial;
    void *scratch;
    size_t bytes;
};
ACarve acarve(void *ws, const ADims &m)
{
    ACarve c;
    char *p = (char *)ws;
    const size_t hw = (size_t)m.h * m.w, HW = (size_t)m.H * m.W;
    ts_carve(p, c.low, (size_t)m.C * hw);
    ts_carve(p, c.low1, hw);
    ts_carve(p, c.U, HW);
    ts_carve(p, c.mm, (size_t)4);
    ts_carve(p, c.pmin, (size_t)SUM_BLOCKS);
    ts_carve(p, c.pmax, (size_t)SUM_BLOCKS);
    ts_carve(p, c.thr, (size_t)4);
    ts_carve(p, c.partial, (size_t)SUM_BLOCKS);
    ts_carve(p, c.adj, (size_t)2 * m.C * HW);
    for (int i = 0; i < 2; i++) { ts_carve(p, c.k[i], HW); ts_carve(p, c.v[i], HW); }
    p = (char *)ts_align_up((size_t)p);
    c.scratch = p;
    p += ts_radix_scratch_bytes(HW) > ts_quantile_scratch_bytes() ? ts_radix_scratch_bytes(HW) : ts_quantile_scratch_bytes(); // (the quantile is a radix select since round 6; the sort scratch is kept as an upper bound)
    c.bytes = (size_t)(p - (char *)ws) + TS_ALIGN;
    return c;
}
} // namespace

size_t ts_aux_loss_workspace_bytes(int C, int H, int W, double scale)
{
    if (C <= 0 || H <= 0 || W <= 0) return TS_ALIGN;
    // one buffer serves the mask (low-resolution planes at `scale`) and the loss / gradient calls (carved for scale 1)
    return std::max(acarve(nullptr, make_adims(C, H, W, scale)).bytes, acarve(nullptr, make_adims(C, H, W, 1.0)).bytes);
}

hipError_t ts_dog_mask(const float *gt, int C, int H, int W, double sigma1, int ksize1, double sigma2, int ksize2, int invert, double scale, void *workspace,
                       float *mask, hipStream_t s)
{
    const ADims m = make_adims(C, H, W, scale);
    const ACarve c = acarve(workspace, m);
    Gauss gs{};
    gs.k1 = ksize1; gs.k2 = ksize2;
    auto fill = [](float *w, int k, double sigma) { // exp(-(x - mean)^2 / (2 sigma^2)), normalised so that outer(w, w) sums to 1 like the 2-D kernel
        double sum = 0.0, tmp[33];
        const double mean = (k - 1) / 2.0;
        for (int i = 0; i < k; i++) { tmp[i] = exp(-(i - mean) * (i - mean) / (2.0 * sigma * sigma)); sum += tmp[i]; }
        for (int i = 0; i < k; i++) w[i] = (float)(tmp[i] / sum);
    };
    fill(gs.w1, ksize1, sigma1);
    fill(gs.w2, ksize2, sigma2);
    const int hw = m.h * m.w, HW = H * W;
    const dim3 lo((unsigned)((hw + 255) / 256)), hi((unsigned)((HW + 255) / 256));
    hipLaunchKernelGGL((aux_downsample_kernel<true>), lo, dim3(256), 0, s, m, gt, c.low1);
    hipLaunchKernelGGL(aux_dog_kernel, lo, dim3(256), 0, s, m, gs, c.low1, c.low);
    const int nb = min(SUM_BLOCKS, (HW + 255) / 256);
    hipLaunchKernelGGL((aux_upsample_kernel<false, true>), dim3((unsigned)nb), dim3(256), 0, s, m, c.low, c.U, (uint32_t *)nullptr, c.pmin, c.pmax);
    hipLaunchKernelGGL(aux_minmax_finish_kernel, dim3(1), dim3(64), 0, s, nb, c.pmin, c.pmax, c.mm);
    hipLaunchKernelGGL(aux_dog_mask_kernel, hi, dim3(256), 0, s, HW, c.U, c.mm, invert, mask);
    return hipGetLastError();
}

hipError_t ts_smoothness_mask(const float *gt, int C, int H, int W, double scale, float quantile, void *workspace, float *mask, hipStream_t s)
{
    const ADims m = make_adims(C, H, W, scale);
    const ACarve c = acarve(workspace, m);
    const int hw = m.h * m.w, HW = H * W;
    const dim3 lo((unsigned)((hw + 255) / 256)), hi((unsigned)((HW + 255) / 256));
    hipLaunchKernelGGL((aux_downsample_kernel<false>), lo, dim3(256), 0, s, m, gt, c.low);
    hipLaunchKernelGGL(aux_scharr_norm_low_kernel, lo, dim3(256), 0, s, m, c.low, c.low1);
    const int nb = min(SUM_BLOCKS, (HW + 255) / 256);
    hipLaunchKernelGGL((aux_upsample_kernel<true, false>), dim3((unsigned)nb), dim3(256), 0, s, m, c.low1, c.U, c.k[0], (float *)nullptr, (float *)nullptr);
    ts_quantile_threshold(c.k[0], (size_t)HW, quantile, c.scratch, c.thr, s); // radix select (select.hip): the norms are >= 0, their bit patterns order like the values
    hipLaunchKernelGGL(aux_below_mask_kernel, hi, dim3(256), 0, s, HW, c.U, c.thr, mask);
    return hipGetLastError();
}

hipError_t ts_masked_l1_forward(const float *img, const float *gt, const float *mask, int C, int H, int W, void *workspace, float *out, hipStream_t s)
{
    const ACarve c = acarve(workspace, make_adims(C, H, W, 1.0));
    const size_t n = (size_t)C * H * W;
    const int nb = (int)std::min<size_t>(SUM_BLOCKS, (n + 255) / 256);
    hipLaunchKernelGGL(aux_masked_l1_sum_kernel, dim3((unsigned)nb), dim3(256), 0, s, C, H * W, img, gt, mask, c.partial);
    hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(64), 0, s, nb, (double)n, c.partial, out);
    return hipGetLastError();
}
hipError_t ts_masked_l1_backward(const float *img, const float *gt, const float *mask, int C, int H, int W, const float *grad_out, float *dimg, hipStream_t s)
{
    const size_t n = (size_t)C * H * W;
    hipLaunchKernelGGL(aux_masked_l1_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, C, H * W, img, gt, mask, grad_out, dimg);
    return hipGetLastError();
}
hipError_t ts_scharr_smoothness_forward(const float *img, const float *mask, int C, int H, int W, void *workspace, float *out, hipStream_t s)
{
    const ACarve c = acarve(workspace, make_adims(C, H, W, 1.0));
    const int HW = H * W, nb = min(SUM_BLOCKS, (HW + 255) / 256);
    hipLaunchKernelGGL(aux_smooth_sum_kernel, dim3((unsigned)nb), dim3(256), 0, s, C, H, W, img, mask, c.partial);
    hipLaunchKernelGGL(aux_finish_kernel, dim3(1), dim3(64), 0, s, nb, (double)HW, c.partial, out);
    return hipGetLastError();
}
hipError_t ts_scharr_smoothness_backward(const float *img, const float *mask, int C, int H, int W, void *workspace, const float *grad_out, float *dimg,
                                         hipStream_t s)
{
    const ACarve c = acarve(workspace, make_adims(C, H, W, 1.0));
    const int HW = H * W;
    hipLaunchKernelGGL(aux_smooth_adj_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, s, C, H, W, img, mask, grad_out, c.adj);
    hipLaunchKernelGGL(aux_smooth_bwd_kernel, dim3((unsigned)(((size_t)C * HW + 255) / 256)), dim3(256), 0, s, C, H, W, c.adj, dimg);
    return hipGetLastError();
}
