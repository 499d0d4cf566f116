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

// ---- round 6: the same separable passes with every LDS access a 128-bit one -------------------------------------------------------------
// Round 1's kernels read one float per tap from LDS (22 + 55 ds_read_b32 per output pixel in the forward): 0.17 / 0.28 of the HBM roofline,
// bound by the LDS pipe.  Now (i) the halo tile is loaded as aligned float4s (columns x0 - 8 .. x0 + 39: 12 float4 per row instead of 42 scalar
// loads; needs W % 4 == 0 and 16-byte aligned planes, otherwise the scalar loader fills the same tile), (ii) a row-pass thread forms FOUR
// adjacent outputs from five ds_read_b128 per image (a sliding window in registers), (iii) the row results are stored TRANSPOSED (column-major,
// stride 28: rows of one column are contiguous), so that a column-pass thread forms four adjacent output rows of one column from four
// ds_read_b128 per array.  Strides 52 / 28 floats make every 8-lane phase of those reads hit 8 x 4 distinct banks.  LDS bytes per output
// pixel: ~400 -> ~165.  Summation order per output is unchanged (taps 0..10 in sequence): same bits as round 1's kernels.
constexpr int LW = 48;            // LDS tile width: TW + 2 x 8 (the 5-pixel halo rounded to float4s)
constexpr int LWP = 52;           // its row stride in floats
constexpr int HS = 28;            // stride of a transposed column: IH = 26 rows, rounded to float4s

template <int NARR, bool VEC>
__device__ __forceinline__ void load_halo_tile(float (*dst)[IH][LWP], const float *const (&src)[NARR], int x0, int y0, int H, int W, int tid)
{
    if (VEC)
    {
        for (int i = tid; i < IH * (LW / 4); i += 256)
        {
            const int r = i / (LW / 4), q = i - r * (LW / 4);
            const int gy = y0 + r - R, gx = x0 - 8 + 4 * q;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W; // W % 4 == 0 and gx % 4 == 0: the four columns are inside or outside together
#pragma unroll
            for (int a = 0; a < NARR; a++)
                *(float4 *)&dst[a][r][4 * q] = in ? *(const float4 *)(src[a] + (size_t)gy * W + gx) : make_float4(0.0f, 0.0f, 0.0f, 0.0f); // zero padding, trainer_utils.py:42-43
        }
    }
    else
    {
        for (int i = tid; i < IH * LW; i += 256)
        {
            const int r = i / LW, c = i - r * LW;
            const int gy = y0 + r - R, gx = x0 - 8 + c;
            const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int a = 0; a < NARR; a++) dst[a][r][c] = in ? src[a][(size_t)gy * W + gx] : 0.0f;
        }
    }
}

// Column pass of one thread: output rows r0 .. r0 + 3 of column c from the transposed row results (rows r0 .. r0 + 13 of that column).
template <int NARR>
__device__ __forceinline__ void column_pass4(const float (*hb)[TW][HS], const Gauss &g, int c, int r0, float (&out)[NARR][4])
{
#pragma unroll
    for (int a = 0; a < NARR; a++)
    {
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) *(float4 *)(v + 4 * q) = *(const float4 *)&hb[a][c][r0 + 4 * q];
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < K; k++) acc = fmaf(g.w[k], v[o + k], acc);
            out[a][o] = acc;
        }
    }
}

template <bool NEED_GRAD, bool VEC>
__global__ void __launch_bounds__(256) ssim_l1_fwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W,
                                                           Gauss g, float *__restrict__ d_mu, float *__restrict__ d_s1,
                                                           float *__restrict__ d_s12, float2 *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float sIG[2][IH][LWP];
    __shared__ __attribute__((aligned(16))) float hb[5][TW][HS];
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *const src[2] = {img + plane, gt + plane};
    load_halo_tile<2, VEC>(sIG, src, x0, y0, H, W, tid);
    __syncthreads();
    if (tid < IH * (TW / 4)) // rows: four adjacent outputs per thread
    {
        const int r = tid % IH, c0 = 4 * (tid / IH);
        float a[20], b[20]; // LDS columns c0 .. c0 + 19; output c0 + o reads taps at columns c0 + o + 3 + k
#pragma unroll
        for (int q = 0; q < 5; q++)
        {
            *(float4 *)(a + 4 * q) = *(const float4 *)&sIG[0][r][c0 + 4 * q];
            *(float4 *)(b + 4 * q) = *(const float4 *)&sIG[1][r][c0 + 4 * q];
        }
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
            for (int k = 0; k < K; k++)
            {
                const float x = a[o + 3 + k], y = b[o + 3 + k], w = g.w[k];
                const float wa = w * x, wb = w * y;
                a0 += wa; a1 += wb; a2 = fmaf(wa, x, a2); a3 = fmaf(wb, y, a3); a4 = fmaf(wa, y, a4);
            }
            hb[0][c0 + o][r] = a0; hb[1][c0 + o][r] = a1; hb[2][c0 + o][r] = a2; hb[3][c0 + o][r] = a3; hb[4][c0 + o][r] = a4;
        }
    }
    __syncthreads();
    float ssim_sum = 0.0f, l1_sum = 0.0f;
    if (tid < TW * (TH / 4)) // columns: four adjacent output rows of one column per thread
    {
        const int c = tid & (TW - 1), r0 = 4 * (tid / TW);
        float cv[5][4];
        column_pass4<5>(hb, g, c, r0, cv);
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            const int r = r0 + o, gy = y0 + r, gx = x0 + c;
            if (gy < H && gx < W)
            {
                // trainer_utils.py:62-75
                const float mu1 = cv[0][o], mu2 = cv[1][o], e11 = cv[2][o], e22 = cv[3][o], e12 = cv[4][o];
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
                const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
                const float iB1 = 1.0f / B1, iB2 = 1.0f / B2;
                const float m = (A1 * A2) * (iB1 * iB2);
                ssim_sum += m;
                l1_sum += fabsf(sIG[0][r + R][c + 8] - sIG[1][r + R][c + 8]); // trainer_utils.py:323-324
                if (NEED_GRAD)
                {
                    // independent variables mu1, e11 = E[I^2], e12 = E[IG] (s1 = e11 - mu1^2, s12 = e12 - mu1 mu2)
                    const float dm_ds1 = -m * iB2;                   // d map / d sigma1^2
                    const float dm_ds12 = 2.0f * A1 * (iB1 * iB2);   // d map / d sigma12
                    const float dm_dmu1 = (2.0f * mu2 * A2 * (iB1 * iB2) - 2.0f * mu1 * m * iB1) // through A1 / B1
                                          - mu2 * dm_ds12 - 2.0f * mu1 * dm_ds1;
                    const size_t ofs = plane + (size_t)gy * W + gx;
                    d_mu[ofs] = dm_dmu1; d_s1[ofs] = dm_ds1; d_s12[ofs] = dm_ds12;
                }
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

template <bool VEC>
__global__ void __launch_bounds__(256) ssim_l1_bwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W,
                                                           Gauss g, const float *__restrict__ d_mu, const float *__restrict__ d_s1,
                                                           const float *__restrict__ d_s12, float k_ssim, float k_l1,
                                                           const float *__restrict__ grad_out, float *__restrict__ dL_dimg)
{
    __shared__ __attribute__((aligned(16))) float sM[3][IH][LWP];
    __shared__ __attribute__((aligned(16))) float hb[3][TW][HS];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t plane = (size_t)blockIdx.z * H * W;
    const float *const src[3] = {d_mu + plane, d_s1 + plane, d_s12 + plane};
    load_halo_tile<3, VEC>(sM, src, x0, y0, H, W, tid);
    __syncthreads();
    if (tid < IH * (TW / 4))
    {
        const int r = tid % IH, c0 = 4 * (tid / IH);
#pragma unroll
        for (int a = 0; a < 3; a++)
        {
            float v[20];
#pragma unroll
            for (int q = 0; q < 5; q++) *(float4 *)(v + 4 * q) = *(const float4 *)&sM[a][r][c0 + 4 * q];
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < K; k++) acc = fmaf(g.w[k], v[o + 3 + k], acc);
                hb[a][c0 + o][r] = acc;
            }
        }
    }
    __syncthreads();
    if (tid < TW * (TH / 4))
    {
        const float go = grad_out ? grad_out[0] : 1.0f;
        const int c = tid & (TW - 1), r0 = 4 * (tid / TW);
        float cv[3][4];
        column_pass4<3>(hb, g, c, r0, cv);
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            const int gy = y0 + r0 + o, gx = x0 + c;
            if (gy < H && gx < W)
            {
                const size_t ofs = plane + (size_t)gy * W + gx;
                const float a = img[ofs], b = gt[ofs];
                const float d = a - b;
                const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); // torch.abs backward: sign, 0 at 0
                dL_dimg[ofs] = go * (k_ssim * (cv[0][o] + 2.0f * a * cv[1][o] + b * cv[2][o]) + k_l1 * sgn);
            }
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
    const bool vec = (W % 4 == 0) && (((size_t)image | (size_t)gt) & 15) == 0; // planes then start on 16-byte boundaries too (H * W * 4 bytes each)
#define TS_SSIM_FWD(G, V) hipLaunchKernelGGL((ssim_l1_fwd_kernel<G, V>), grid, dim3(256), 0, s, image, gt, H, W, gauss(), c.d_mu, c.d_s1, c.d_s12, c.partial)
    if (need_grad) { if (vec) TS_SSIM_FWD(true, true); else TS_SSIM_FWD(true, false); }
    else { if (vec) TS_SSIM_FWD(false, true); else TS_SSIM_FWD(false, false); }
#undef TS_SSIM_FWD
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
    const bool vec = (W % 4 == 0) && (((size_t)c.d_mu | (size_t)c.d_s1 | (size_t)c.d_s12) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(ssim_l1_bwd_kernel<true>, grid, dim3(256), 0, s, image, gt, H, W, gauss(), c.d_mu, c.d_s1, c.d_s12,
                           (float)(-(double)w_ssim * inv_n), (float)((double)w_l1 * inv_n), grad_out, dL_dimage);
    else
        hipLaunchKernelGGL(ssim_l1_bwd_kernel<false>, grid, dim3(256), 0, s, image, gt, H, W, gauss(), c.d_mu, c.d_s1, c.d_s12,
                           (float)(-(double)w_ssim * inv_n), (float)((double)w_l1 * inv_n), grad_out, dL_dimage);
    return hipGetLastError();
}
