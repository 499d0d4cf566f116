// photometric.hip -- fused L1 + SSIM loss, forward and backward (include/ts_loss.h).
//
// Reference behaviour (src/diff_recon/trainers/trainer_utils.py:9-103, 323-324): five depthwise 11x11 Gaussian
// convolutions with ZERO padding (mu1, mu2, E[I^2], E[G^2], E[IG]), the SSIM map, its mean, and mean|I - G|; autograd
// for the backward.  Here, per 32x16 output tile and channel:
//   forward : load the 42x26 halo tile of both images into LDS once, separable convolution of the five products
//             (rows, then columns) in LDS, SSIM value and its three partial derivatives per pixel, block-reduced sums;
//   backward: dL/dI = conv(dmap/dmu1) + 2 I conv(dmap/dE[I^2]) + G conv(dmap/dE[IG]) (the Gaussian is symmetric and
//             zero padding is self-adjoint), again one halo load + separable pass in LDS, plus the L1 sign term.
// Both are HBM/L2-bound: forward reads 2 and writes 3 floats per element, backward reads 5 and writes 1.
// Final sums are formed in double by a one-block kernel, so the loss value is run-to-run deterministic.
#include "../../include/ts_loss.h"
#include "ts2d_common.h"

namespace
{
constexpr int R = 5, K = 11;           // window radius / size
constexpr int TW = 32, TH = 16;        // output tile
constexpr int IW = TW + 2 * R;         // 42 input columns
constexpr int IH = TH + 2 * R;         // 26 input rows
constexpr int IWP = IW + 2;            // padded LDS row stride
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Gauss { float w[K]; };

Gauss make_gauss()
{
    // trainer_utils.py:17-29 with kernel_size 11, sigma 1.5: exp(-(dx^2 + dy^2) / (2 sigma^2)) / sum  ==  outer product
    // of the normalised 1-D Gaussian
    Gauss g;
    double s = 0.0, v[K];
    for (int i = 0; i < K; i++) { v[i] = exp(-(double)((i - R) * (i - R)) / (2.0 * 1.5 * 1.5)); s += v[i]; }
    for (int i = 0; i < K; i++) g.w[i] = (float)(v[i] / s);
    return g;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <bool NEED_GRAD>
__global__ void __launch_bounds__(256) ssim_l1_fwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W,
                                                           Gauss g, float *__restrict__ d_mu, float *__restrict__ d_s1,
                                                           float *__restrict__ d_s12, float2 *__restrict__ partial)
{
    __shared__ float sI[IH][IWP], sG[IH][IWP];
    __shared__ float hb[5][IH][TW];
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *I = img + plane, *G = gt + plane;

    for (int i = tid; i < IH * IW; i += 256)
    {
        const int r = i / IW, c = i - r * IW;
        const int gy = y0 + r - R, gx = x0 + c - R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        sI[r][c] = in ? I[(size_t)gy * W + gx] : 0.0f; // zero padding, trainer_utils.py:42-43
        sG[r][c] = in ? G[(size_t)gy * W + gx] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < IH * TW; i += 256) // rows
    {
        const int r = i / TW, c = i - r * TW;
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            const float a = sI[r][c + k], b = sG[r][c + k], w = g.w[k];
            const float wa = w * a, wb = w * b;
            a0 += wa; a1 += wb; a2 = fmaf(wa, a, a2); a3 = fmaf(wb, b, a3); a4 = fmaf(wa, b, a4);
        }
        hb[0][r][c] = a0; hb[1][r][c] = a1; hb[2][r][c] = a2; hb[3][r][c] = a3; hb[4][r][c] = a4;
    }
    __syncthreads();
    float ssim_sum = 0.0f, l1_sum = 0.0f;
    const int c = tid & 31;
#pragma unroll
    for (int half = 0; half < 2; half++) // columns: two output pixels per thread
    {
        const int r = (tid >> 5) + half * 8;
        const int gy = y0 + r, gx = x0 + c;
        float mu1 = 0, mu2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            const float w = g.w[k];
            mu1 = fmaf(w, hb[0][r + k][c], mu1); mu2 = fmaf(w, hb[1][r + k][c], mu2);
            e11 = fmaf(w, hb[2][r + k][c], e11); e22 = fmaf(w, hb[3][r + k][c], e22);
            e12 = fmaf(w, hb[4][r + k][c], e12);
        }
        if (gy < H && gx < W)
        {
            // trainer_utils.py:62-75
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
            const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
            const float iB1 = 1.0f / B1, iB2 = 1.0f / B2;
            const float m = (A1 * A2) * (iB1 * iB2);
            ssim_sum += m;
            l1_sum += fabsf(sI[r + R][c + R] - sG[r + R][c + R]); // trainer_utils.py:323-324
            if (NEED_GRAD)
            {
                // independent variables mu1, e11 = E[I^2], e12 = E[IG] (s1 = e11 - mu1^2, s12 = e12 - mu1 mu2)
                const float dm_ds1 = -m * iB2;                   // d map / d sigma1^2
                const float dm_ds12 = 2.0f * A1 * (iB1 * iB2);   // d map / d sigma12
                const float dm_dmu1 = (2.0f * mu2 * A2 * (iB1 * iB2) - 2.0f * mu1 * m * iB1) // through A1 / B1
                                      - mu2 * dm_ds12 - 2.0f * mu1 * dm_ds1;
                const size_t o = plane + (size_t)gy * W + gx;
                d_mu[o] = dm_dmu1; d_s1[o] = dm_ds1; d_s12[o] = dm_ds12;
            }
        }
    }
    ssim_sum = wave_sum(ssim_sum);
    l1_sum = wave_sum(l1_sum);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ssim_sum; red[1][tid >> 6] = l1_sum; }
    __syncthreads();
    if (tid == 0)
    {
        const int b = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[b] = make_float2(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) loss_finish_kernel(const float2 *__restrict__ partial, int nblocks, double inv_n, float w_l1,
                                                           float w_ssim, float *__restrict__ out)
{
    __shared__ double rs[256], rl[256];
    double s = 0.0, l = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { s += (double)partial[i].x; l += (double)partial[i].y; }
    rs[threadIdx.x] = s; rl[threadIdx.x] = l;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1)
    {
        if ((int)threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rl[threadIdx.x] += rl[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const double l1 = rl[0] * inv_n, ssim_loss = 1.0 - rs[0] * inv_n; // trainer_utils.py:76,103
        out[0] = (float)((double)w_l1 * l1 + (double)w_ssim * ssim_loss);
        out[1] = (float)l1;
        out[2] = (float)ssim_loss;
    }
}

__global__ void __launch_bounds__(256) ssim_l1_bwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W,
                                                           Gauss g, const float *__restrict__ d_mu, const float *__restrict__ d_s1,
                                                           const float *__restrict__ d_s12, float k_ssim, float k_l1,
                                                           const float *__restrict__ grad_out, float *__restrict__ dL_dimg)
{
    __shared__ float sM[3][IH][IWP];
    __shared__ float hb[3][IH][TW];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *M0 = d_mu + plane, *M1 = d_s1 + plane, *M2 = d_s12 + plane;
    for (int i = tid; i < IH * IW; i += 256)
    {
        const int r = i / IW, c = i - r * IW;
        const int gy = y0 + r - R, gx = x0 + c - R;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const size_t o = (size_t)gy * W + gx;
        sM[0][r][c] = in ? M0[o] : 0.0f;
        sM[1][r][c] = in ? M1[o] : 0.0f;
        sM[2][r][c] = in ? M2[o] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < IH * TW; i += 256)
    {
        const int r = i / TW, c = i - r * TW;
        float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            const float w = g.w[k];
            a0 = fmaf(w, sM[0][r][c + k], a0); a1 = fmaf(w, sM[1][r][c + k], a1); a2 = fmaf(w, sM[2][r][c + k], a2);
        }
        hb[0][r][c] = a0; hb[1][r][c] = a1; hb[2][r][c] = a2;
    }
    __syncthreads();
    const float go = grad_out ? grad_out[0] : 1.0f;
    const int c = tid & 31;
#pragma unroll
    for (int half = 0; half < 2; half++)
    {
        const int r = (tid >> 5) + half * 8;
        const int gy = y0 + r, gx = x0 + c;
        float v0 = 0, v1 = 0, v2 = 0;
#pragma unroll
        for (int k = 0; k < K; k++)
        {
            const float w = g.w[k];
            v0 = fmaf(w, hb[0][r + k][c], v0); v1 = fmaf(w, hb[1][r + k][c], v1); v2 = fmaf(w, hb[2][r + k][c], v2);
        }
        if (gy < H && gx < W)
        {
            const size_t o = plane + (size_t)gy * W + gx;
            const float a = img[o], b = gt[o];
            const float d = a - b;
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); // torch.abs backward: sign, 0 at 0
            dL_dimg[o] = go * (k_ssim * (v0 + 2.0f * a * v1 + b * v2) + k_l1 * sgn);
        }
    }
}

struct Carve
{
    float *d_mu, *d_s1, *d_s12;
    float2 *partial;
    int nblocks;
    size_t bytes;
};

Carve carve(void *ws, int C, int H, int W)
{
    Carve c;
    const size_t n = (size_t)C * H * W;
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, C);
    c.nblocks = (int)(grid.x * grid.y * grid.z);
    char *p = (char *)ts_align_up((size_t)ws);
    c.d_mu = (float *)p; p += ts_align_up(n * 4);
    c.d_s1 = (float *)p; p += ts_align_up(n * 4);
    c.d_s12 = (float *)p; p += ts_align_up(n * 4);
    c.partial = (float2 *)p; p += ts_align_up((size_t)c.nblocks * 8);
    c.bytes = (size_t)(p - (char *)ws);
    return c;
}

const Gauss &gauss()
{
    static const Gauss g = make_gauss();
    return g;
}
} // namespace

size_t ts_loss_workspace_bytes(int C, int H, int W)
{
    if (C <= 0 || H <= 0 || W <= 0) return TS_ALIGN;
    return carve(nullptr, C, H, W).bytes + TS_ALIGN;
}

hipError_t ts_loss_forward(const float *image, const float *gt, int C, int H, int W, float w_l1, float w_ssim, bool need_grad,
                           void *workspace, float *out, hipStream_t s)
{
    const Carve c = carve(workspace, C, H, W);
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, C);
    if (need_grad)
        hipLaunchKernelGGL(ssim_l1_fwd_kernel<true>, grid, dim3(256), 0, s, image, gt, H, W, gauss(), c.d_mu, c.d_s1, c.d_s12, c.partial);
    else
        hipLaunchKernelGGL(ssim_l1_fwd_kernel<false>, grid, dim3(256), 0, s, image, gt, H, W, gauss(), c.d_mu, c.d_s1, c.d_s12, c.partial);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, c.partial, c.nblocks, 1.0 / ((double)C * H * W), w_l1, w_ssim, out);
    return hipGetLastError();
}

hipError_t ts_loss_backward(const float *image, const float *gt, int C, int H, int W, float w_l1, float w_ssim, const void *workspace,
                            const float *grad_out, float *dL_dimage, hipStream_t s)
{
    const Carve c = carve(const_cast<void *>(workspace), C, H, W);
    const dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, C);
    const double inv_n = 1.0 / ((double)C * H * W);
    // d(1 - mean(map)) = -1/N per map element; d mean|I - G| = sign / N
    hipLaunchKernelGGL(ssim_l1_bwd_kernel, grid, dim3(256), 0, s, image, gt, H, W, gauss(), c.d_mu, c.d_s1, c.d_s12,
                       (float)(-(double)w_ssim * inv_n), (float)((double)w_l1 * inv_n), grad_out, dL_dimage);
    return hipGetLastError();
}
