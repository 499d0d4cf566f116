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
constexpr int IH = TH + 2 * R;         // 26 input rows
constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;

struct Gauss { float w[K]; };
// The same eleven weights as compile-time constants for the kernels (make_gauss() below evaluates to exactly these floats; `gauss_matches_constants`
// checks it once per process): as kernel arguments they occupied 11 scalar registers of every wave and spilled.
struct GaussK
{
    static constexpr float w[K] = {0x1.0d956cp-10f, 0x1.f1fe02p-8f, 0x1.26eb18p-5f, 0x1.bff0fep-4f, 0x1.b43c4p-3f, 0x1.10656p-2f,
                                   0x1.b43c4p-3f, 0x1.bff0fep-4f, 0x1.26eb18p-5f, 0x1.f1fe02p-8f, 0x1.0d956cp-10f};
};

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

// ---- round 6: sliding windows in registers, conflict-free LDS accesses, XCD-aware tiles --------------------------------------------------
// Round 1's kernels read one float per tap from LDS (22 + 55 ds_read_b32 per output pixel in the forward): 0.17 / 0.28 of the HBM roofline.
// Now (i) the halo tile is loaded as aligned float4s (columns x0 - 8 .. x0 + 39: 12 float4 per row instead of 42 scalar loads; needs W % 4 == 0 and
// 16-byte aligned planes, otherwise the scalar loader fills the same tile); (ii) a row-pass thread forms FOUR adjacent outputs from five
// ds_read_b128 per image (a sliding window in registers) and stores them with one ds_write_b128 per array -- consecutive lanes take consecutive
// 16-byte pieces of a row, so every access is a contiguous 128 bytes per 8 lanes; (iii) a column-pass thread forms four adjacent output ROWS of one
// column from 14 row-contiguous ds_read_b32 per array (consecutive lanes = consecutive columns).  A first version stored the row results
// transposed to read them back as b128 columns: the counters (profiles/r06_loss_pmc.txt) showed 63 % of its LDS cycles were bank conflicts and no
// gain over round 1.  Summation order per output is unchanged (taps 0..10 in sequence): same bits as round 1's kernels.
#ifndef TS_SSIM_WG_PER_XCD
#define TS_SSIM_WG_PER_XCD (1 << 24) /* workgroups per XCD: one per tile.  160 (= 32 compute units x 5 resident workgroups: persistent, the next tile's
                                        loads in flight under the current tile's passes) was measured: SLOWER, 0.104 vs 0.089 ms forward -- the prefetch
                                        registers cost a wave per SIMD and the kernel is not waiting for memory (profiles/r06_loss_pmc.txt) */
#endif
constexpr int LW = 48;            // LDS tile width: TW + 2 x 8 (the 5-pixel halo rounded to float4s)
constexpr int LWP = 52;           // its row stride in floats
constexpr int HS = TW + 4;        // row stride of the row-pass results (floats): 16-byte aligned rows, consecutive rows 4 banks apart

// Tiles -> workgroups, XCD-aware and persistent.  Workgroups go to the 8 XCDs round-robin and every XCD has its own L2: XCD k takes the k-th eighth
// of the tiles in (plane, y, x) order -- a band of tile rows, so that the tiles that share halo columns and halo rows meet in ONE L2 -- and the
// workgroups of an XCD walk through that eighth together (workgroup j of the XCD takes tiles j, j + n, j + 2 n, ...).  A workgroup issues the global
// loads of its NEXT tile into registers before it computes the current one: the two barriers and the dependent LDS phases of a tile otherwise
// leave every wave waiting for memory at the top of each tile (3.9 waves per SIMD, VALU 31 % busy, profiles/r06_loss_pmc.txt).
struct TileWalk
{
    int nb, per, xcd, j, stride, gx, gy;
    __device__ __forceinline__ TileWalk(int gx_, int gy_, int planes) : gx(gx_), gy(gy_)
    {
        nb = gx * gy * planes; per = (nb + 7) >> 3; xcd = (int)(blockIdx.x & 7); j = (int)(blockIdx.x >> 3); stride = (int)(gridDim.x >> 3);
    }
    __device__ __forceinline__ bool valid(int jj) const { return jj < per && xcd * per + jj < nb; }
    __device__ __forceinline__ void decode(int jj, int &tx, int &ty, int &tz) const
    {
        const int t = xcd * per + jj;
        tx = t % gx; ty = (t / gx) % gy; tz = t / (gx * gy);
    }
};
constexpr bool PERSISTENT = TS_SSIM_WG_PER_XCD < (1 << 24); // compile time: the one-tile-per-workgroup form carries no prefetch registers and no loop
constexpr int NPRE = (IH * 12 + 255) / 256; // float4 pieces of a halo tile per thread and array (IH rows x 12 float4)

template <int NARR>
__device__ __forceinline__ void prefetch_issue(float4 (&pre)[NARR][NPRE], const float *const (&src)[NARR], int x0, int y0, int H, int W, int tid)
{
#pragma unroll
    for (int q = 0; q < NPRE; q++)
    {
        const int i = tid + 256 * q;
        const int r = i / 12, qq = i - r * 12;
        const int gy = y0 + r - R, gx = x0 - 8 + 4 * qq;
        const bool in = i < IH * 12 && gy >= 0 && gy < H && gx >= 0 && gx < W; // W % 4 == 0 and gx % 4 == 0: four columns inside or outside together
#pragma unroll
        for (int a = 0; a < NARR; a++) pre[a][q] = in ? *(const float4 *)(src[a] + (size_t)gy * W + gx) : make_float4(0.0f, 0.0f, 0.0f, 0.0f); // zero padding, trainer_utils.py:42-43
    }
}
template <int NARR>
__device__ __forceinline__ void prefetch_commit(float (*dst)[IH][LWP], const float4 (&pre)[NARR][NPRE], int tid)
{
#pragma unroll
    for (int q = 0; q < NPRE; q++)
    {
        const int i = tid + 256 * q;
        const int r = i / 12, qq = i - r * 12;
        if (i < IH * 12)
#pragma unroll
            for (int a = 0; a < NARR; a++) *(float4 *)&dst[a][r][4 * qq] = pre[a][q];
    }
}

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

// Column pass of one thread: output rows r0 .. r0 + 3 of column c from rows r0 .. r0 + 13 of the row results.
template <int NARR>
__device__ __forceinline__ void column_pass4(const float (*hb)[IH][HS], int c, int r0, float (&out)[NARR][4])
{
#pragma unroll
    for (int a = 0; a < NARR; a++)
    {
        float v[14];
#pragma unroll
        for (int q = 0; q < 14; q++) v[q] = hb[a][r0 + q][c];
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < K; k++) acc = fmaf(GaussK::w[k], v[o + k], acc);
            out[a][o] = acc;
        }
    }
}

template <bool NEED_GRAD, bool VEC>
__global__ void __launch_bounds__(256) ssim_l1_fwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W,
                                                           float *__restrict__ d_mu, float *__restrict__ d_s1,
                                                           float *__restrict__ d_s12, float2 *__restrict__ partial, int gx, int gy, int planes)
{
    __shared__ __attribute__((aligned(16))) float sIG[2][IH][LWP];
    __shared__ __attribute__((aligned(16))) float hb[5][IH][HS];
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    TileWalk walk(gx, gy, planes);
    bool have = walk.valid(walk.j);
    int tx = 0, ty = 0, tz = 0;
    float4 pre[(PERSISTENT && VEC) ? 2 : 1][NPRE];
    if (!have) return; // the grid is padded to a multiple of 8
    walk.decode(walk.j, tx, ty, tz);
    if (PERSISTENT && VEC)
    {
        const float *const src0[2] = {img + (size_t)tz * H * W, gt + (size_t)tz * H * W};
        prefetch_issue<2>((float4(&)[2][NPRE])pre, src0, tx * TW, ty * TH, H, W, tid);
    }
    do
    {
    const int x0 = tx * TW, y0 = ty * TH;
    const size_t plane = (size_t)tz * H * W;
    const int ctx = tx, cty = ty, ctz = tz;
    if (PERSISTENT && VEC) prefetch_commit<2>(sIG, (const float4(&)[2][NPRE])pre, tid);
    else
    {
        const float *const src[2] = {img + plane, gt + plane};
        load_halo_tile<2, VEC>(sIG, src, x0, y0, H, W, tid);
    }
    __syncthreads();
    if (PERSISTENT)
    {
        walk.j += walk.stride;
        have = walk.valid(walk.j);
        if (have)
        {
            walk.decode(walk.j, tx, ty, tz);
            const float *const srcn[2] = {img + (size_t)tz * H * W, gt + (size_t)tz * H * W};
            if (VEC) prefetch_issue<2>((float4(&)[2][NPRE])pre, srcn, tx * TW, ty * TH, H, W, tid); // in flight while this tile is computed
        }
    }
    if (tid < IH * (TW / 4)) // rows: four adjacent outputs per thread
    {
        const int r = tid >> 3, c0 = 4 * (tid & 7);
        float a[20], b[20]; // LDS columns c0 .. c0 + 19; output c0 + o reads taps at columns c0 + o + 3 + k
#pragma unroll
        for (int q = 0; q < 5; q++)
        {
            *(float4 *)(a + 4 * q) = *(const float4 *)&sIG[0][r][c0 + 4 * q];
            *(float4 *)(b + 4 * q) = *(const float4 *)&sIG[1][r][c0 + 4 * q];
        }
        float res[5][4]; // (forming the three products once per input element and five FMAs per tap was measured: MORE instructions after
                         //  register allocation and one wave per SIMD less, 0.097 vs 0.089 ms)
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
#pragma unroll
            for (int k = 0; k < K; k++)
            {
                const float x = a[o + 3 + k], y = b[o + 3 + k], w = GaussK::w[k];
                const float wa = w * x, wb = w * y;
                a0 += wa; a1 += wb; a2 = fmaf(wa, x, a2); a3 = fmaf(wb, y, a3); a4 = fmaf(wa, y, a4);
            }
            res[0][o] = a0; res[1][o] = a1; res[2][o] = a2; res[3][o] = a3; res[4][o] = a4;
        }
#pragma unroll
        for (int q = 0; q < 5; q++) *(float4 *)&hb[q][r][c0] = make_float4(res[q][0], res[q][1], res[q][2], res[q][3]);
    }
    __syncthreads();
    float ssim_sum = 0.0f, l1_sum = 0.0f;
    if (tid < TW * (TH / 4)) // columns: four adjacent output rows of one column per thread
    {
        const int c = tid & (TW - 1), r0 = 4 * (tid / TW);
        float cv[5][4];
        column_pass4<5>(hb, c, r0, cv);
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            const int r = r0 + o, py = y0 + r, pxx = x0 + c;
            if (py < H && pxx < W)
            {
                // trainer_utils.py:62-75
                const float mu1 = cv[0][o], mu2 = cv[1][o], e11 = cv[2][o], e22 = cv[3][o], e12 = cv[4][o];
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float s1 = e11 - mu1_sq, s2 = e22 - mu2_sq, s12 = e12 - mu12;
                const float A1 = 2.0f * mu12 + C1, A2 = 2.0f * s12 + C2, B1 = mu1_sq + mu2_sq + C1, B2 = s1 + s2 + C2;
                const float iB1 = 1.0f / B1, iB2 = 1.0f / B2; // (IEEE divisions, like the reference: ~25 of the ~300 instructions per pixel)
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
                    const size_t ofs = plane + (size_t)py * W + pxx;
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
        const int b = (ctz * gy + cty) * gx + ctx;
        partial[b] = make_float2(red[0][0] + red[0][1] + red[0][2] + red[0][3], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
    // (the barrier above also separates this tile's last reads of sIG / hb from the next tile's commit)
    } while (PERSISTENT && have);
}

// 1024 threads: at 1080p there are 12 240 partial pairs, and 256 threads walking 48 of them each + an 8-level LDS tree of doubles took 15 us
// (profiles/r06_train_step_kernels.txt); this is 12 each, a wave shuffle network and one LDS exchange between the 16 waves.
__global__ void __launch_bounds__(1024) loss_finish_kernel(const float2 *__restrict__ partial, int nblocks, double inv_n, float w_l1,
                                                            float w_ssim, float *__restrict__ out)
{
    __shared__ double rs[16], rl[16];
    double s = 0.0, l = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) { const float2 p = partial[i]; s += (double)p.x; l += (double)p.y; }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); l += __shfl_xor(l, o); }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rl[threadIdx.x >> 6] = l; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        double ts = 0.0, tl = 0.0;
        for (int w = 0; w < 16; ++w) { ts += rs[w]; tl += rl[w]; }
        const double l1 = tl * inv_n, ssim_loss = 1.0 - ts * inv_n; // trainer_utils.py:76,103
        out[0] = (float)((double)w_l1 * l1 + (double)w_ssim * ssim_loss);
        out[1] = (float)l1;
        out[2] = (float)ssim_loss;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) ssim_l1_bwd_kernel(const float *__restrict__ img, const float *__restrict__ gt, int H, int W,
                                                           const float *__restrict__ d_mu, const float *__restrict__ d_s1,
                                                           const float *__restrict__ d_s12, float k_ssim, float k_l1,
                                                           const float *__restrict__ grad_out, float *__restrict__ dL_dimg, int gx, int gy, int planes)
{
    __shared__ __attribute__((aligned(16))) float sM[3][IH][LWP];
    __shared__ __attribute__((aligned(16))) float hb[3][IH][HS];
    const int tid = threadIdx.x;
    TileWalk walk(gx, gy, planes);
    bool have = walk.valid(walk.j);
    int tx = 0, ty = 0, tz = 0;
    float4 pre[(PERSISTENT && VEC) ? 3 : 1][NPRE];
    if (!have) return;
    walk.decode(walk.j, tx, ty, tz);
    if (PERSISTENT && VEC)
    {
        const size_t pl = (size_t)tz * H * W;
        const float *const src0[3] = {d_mu + pl, d_s1 + pl, d_s12 + pl};
        prefetch_issue<3>((float4(&)[3][NPRE])pre, src0, tx * TW, ty * TH, H, W, tid);
    }
    do
    {
    const int x0 = tx * TW, y0 = ty * TH;
    const size_t plane = (size_t)tz * H * W;
    if (PERSISTENT && VEC) prefetch_commit<3>(sM, (const float4(&)[3][NPRE])pre, tid);
    else
    {
        const float *const src[3] = {d_mu + plane, d_s1 + plane, d_s12 + plane};
        load_halo_tile<3, VEC>(sM, src, x0, y0, H, W, tid);
    }
    __syncthreads();
    if (PERSISTENT)
    {
        walk.j += walk.stride;
        have = walk.valid(walk.j);
        if (have)
        {
            walk.decode(walk.j, tx, ty, tz);
            const size_t pl = (size_t)tz * H * W;
            const float *const srcn[3] = {d_mu + pl, d_s1 + pl, d_s12 + pl};
            if (VEC) prefetch_issue<3>((float4(&)[3][NPRE])pre, srcn, tx * TW, ty * TH, H, W, tid);
        }
    }
    if (tid < IH * (TW / 4))
    {
        const int r = tid >> 3, c0 = 4 * (tid & 7);
#pragma unroll
        for (int a = 0; a < 3; a++)
        {
            float v[20], res[4];
#pragma unroll
            for (int q = 0; q < 5; q++) *(float4 *)(v + 4 * q) = *(const float4 *)&sM[a][r][c0 + 4 * q];
#pragma unroll
            for (int o = 0; o < 4; o++)
            {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < K; k++) acc = fmaf(GaussK::w[k], v[o + 3 + k], acc);
                res[o] = acc;
            }
            *(float4 *)&hb[a][r][c0] = make_float4(res[0], res[1], res[2], res[3]);
        }
    }
    __syncthreads();
    if (tid < TW * (TH / 4))
    {
        const float go = grad_out ? grad_out[0] : 1.0f;
        const int c = tid & (TW - 1), r0 = 4 * (tid / TW);
        float cv[3][4];
        column_pass4<3>(hb, c, r0, cv);
#pragma unroll
        for (int o = 0; o < 4; o++)
        {
            const int py = y0 + r0 + o, pxx = x0 + c;
            if (py < H && pxx < W)
            {
                const size_t ofs = plane + (size_t)py * W + pxx;
                const float a = img[ofs], b = gt[ofs];
                const float d = a - b;
                const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); // torch.abs backward: sign, 0 at 0
                dL_dimg[ofs] = go * (k_ssim * (cv[0][o] + 2.0f * a * cv[1][o] + b * cv[2][o]) + k_l1 * sgn);
            }
        }
    }
    if (PERSISTENT) __syncthreads(); // this tile's last reads of hb / sM before the next tile's commit
    } while (PERSISTENT && have);
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

bool gauss_matches_constants() // the formula of trainer_utils.py:17-29 against the constants the kernels use
{
    static const bool ok = [] {
        const Gauss g = make_gauss();
        for (int i = 0; i < K; i++)
            if (g.w[i] != GaussK::w[i]) return false;
        return true;
    }();
    return ok;
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
    if (!gauss_matches_constants()) return hipErrorAssert;
    const Carve c = carve(workspace, C, H, W);
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int per = (gx * gy * C + 7) / 8;
    const dim3 grid((unsigned)(8 * (per < TS_SSIM_WG_PER_XCD ? per : TS_SSIM_WG_PER_XCD))); // persistent workgroups, XCD-aware (TileWalk)
    const bool vec = (W % 4 == 0) && (((size_t)image | (size_t)gt) & 15) == 0; // planes then start on 16-byte boundaries too (H * W * 4 bytes each)
#define TS_SSIM_FWD(G, V) hipLaunchKernelGGL((ssim_l1_fwd_kernel<G, V>), grid, dim3(256), 0, s, image, gt, H, W, c.d_mu, c.d_s1, c.d_s12, c.partial, gx, gy, C)
    if (need_grad) { if (vec) TS_SSIM_FWD(true, true); else TS_SSIM_FWD(true, false); }
    else { if (vec) TS_SSIM_FWD(false, true); else TS_SSIM_FWD(false, false); }
#undef TS_SSIM_FWD
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1024), 0, s, c.partial, c.nblocks, 1.0 / ((double)C * H * W), w_l1, w_ssim, out);
    return hipGetLastError();
}

hipError_t ts_loss_backward(const float *image, const float *gt, int C, int H, int W, float w_l1, float w_ssim, const void *workspace,
                            const float *grad_out, float *dL_dimage, hipStream_t s)
{
    const Carve c = carve(const_cast<void *>(workspace), C, H, W);
    const int gx = (W + TW - 1) / TW, gy = (H + TH - 1) / TH;
    const int per = (gx * gy * C + 7) / 8;
    const dim3 grid((unsigned)(8 * (per < TS_SSIM_WG_PER_XCD ? per : TS_SSIM_WG_PER_XCD)));
    const double inv_n = 1.0 / ((double)C * H * W);
    // d(1 - mean(map)) = -1/N per map element; d mean|I - G| = sign / N
    const bool vec = (W % 4 == 0) && (((size_t)c.d_mu | (size_t)c.d_s1 | (size_t)c.d_s12) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(ssim_l1_bwd_kernel<true>, grid, dim3(256), 0, s, image, gt, H, W, c.d_mu, c.d_s1, c.d_s12,
                           (float)(-(double)w_ssim * inv_n), (float)((double)w_l1 * inv_n), grad_out, dL_dimage, gx, gy, C);
    else
        hipLaunchKernelGGL(ssim_l1_bwd_kernel<false>, grid, dim3(256), 0, s, image, gt, H, W, c.d_mu, c.d_s1, c.d_s12,
                           (float)(-(double)w_ssim * inv_n), (float)((double)w_l1 * inv_n), grad_out, dL_dimage, gx, gy, C);
    return hipGetLastError();
}
