// depth_normal.hip -- fused depth / normal consistency loss, forward and backward (include/ts_loss.h).
//
// The producer of the rasterizer's dL_dout_depth / dL_dout_normal in the reference's *_VanillaTS_mesh.yaml configurations
// (geometry_loss.w_geometry = 0.05, scale_factor = 0.5): DepthNormalLoss, src/diff_recon/trainers/trainer_utils.py:204-257, called by
// VanillaTS_trainer.py:84.  Reference behaviour, eager torch (about 40 kernels forward, more backward, one full sort for the quantile):
//   d       = F.interpolate(depth, scale_factor, bilinear, align_corners=False)                        (:214-216)
//   (gx,gy) = Scharr(d) with zero padding, kernels [[-3,0,3],[-10,0,10],[-3,0,3]] / 32 and its transpose (:151-178, :218)
//   Dx, Dy  = gx / d, gy / d;  n = (w Dx / (2 tan_fovx), h Dy / (2 tan_fovy), -(1 + (x - w/2 + .5) Dx + (y - h/2 + .5) Dy))  (:219-229)
//   N       = F.interpolate(n, size = (H, W), bilinear);  Dn = N / |N|                                  (:231-233)
//   G       = F.interpolate(|(gx, gy)|, size = (H, W));  mask = G < quantile(G, 0.9)                    (:237-242, no gradient)
//   loss    = mean((1 - <normalize(normal, eps = 1e-8), Dn>) * mask)                                    (:255-257)
// Here: three elementwise kernels forward (low-resolution depth; Scharr + raw normal; full-resolution terms), the quantile from the
// library's own radix sort of the G values (positive floats sort as integers), a deterministic two-stage sum; four kernels backward
// (full-resolution adjoints; gather-form adjoint of the up-sampling; Scharr adjoint; gather-form adjoint of the down-sampling).  No
// atomics: every adjoint is a gather, so results are run-to-run identical.  All of it is HBM-bound.
#include "../../include/ts_loss.h"
#include "ts2d_common.h"
#include "ts2d_imgops.h"
#include "ts2d_select.h"

namespace
{
struct Dims { int H, W, h, w; float r_down, r_up_y, r_up_x, A, B; }; // A = w / (2 tan_fovx), B = h / (2 tan_fovy)

__global__ void __launch_bounds__(256) dn_downsample_kernel(Dims m, const float *__restrict__ depth, float *__restrict__ d, uint32_t *__restrict__ sel_words, int n_sel_words)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n_sel_words) sel_words[k] = 0u; // the quantile's select state (ts2d_select.h), cleared by the step's first launch instead of one of its own
    if (k >= m.h * m.w) return;
    const int i = k / m.w, j = k - i * m.w;
    if (m.h == m.H && m.w == m.W) { d[k] = depth[k]; return; }
    const Tap ty = tap_of(i, m.r_down, m.H), tx = tap_of(j, m.r_down, m.W);
    const float *r0 = depth + (size_t)ty.i0 * m.W, *r1 = depth + (size_t)ty.i1 * m.W;
    d[k] = ty.l0 * (tx.l0 * r0[tx.i0] + tx.l1 * r0[tx.i1]) + ty.l1 * (tx.l0 * r1[tx.i0] + tx.l1 * r1[tx.i1]);
}

// raw normal (3 planes) and gradient norm at low resolution
__global__ void __launch_bounds__(256) dn_lowres_kernel(Dims m, const float *__restrict__ d, float *__restrict__ nraw, float *__restrict__ gnorm)
{
    const int k = blockIdx.x * 256 + threadIdx.x, hw = m.h * m.w;
    if (k >= hw) return;
    const int i = k / m.w, j = k - i * m.w;
    float gx, gy;
    scharr(d, i, j, m.h, m.w, gx, gy);
    const float dv = d[k], Dx = gx / dv, Dy = gy / dv;
    nraw[k] = m.A * Dx;
    nraw[hw + k] = m.B * Dy;
    nraw[2 * hw + k] = -(1.0f + ((float)j - 0.5f * (float)m.w + 0.5f) * Dx + ((float)i - 0.5f * (float)m.h + 0.5f) * Dy);
    gnorm[k] = sqrtf(gx * gx + gy * gy);
}

__device__ __forceinline__ void fullres_normal(const Dims &m, const float *nraw, int y, int x, float &Nx, float &Ny, float &Nz)
{
    const int hw = m.h * m.w;
    if (m.h == m.H && m.w == m.W)
    {
        const size_t k = (size_t)y * m.W + x;
        Nx = nraw[k]; Ny = nraw[hw + k]; Nz = nraw[2 * hw + k];
        return;
    }
    const Tap ty = tap_of(y, m.r_up_y, m.h), tx = tap_of(x, m.r_up_x, m.w);
    Nx = bilerp(nraw, m.w, ty, tx);
    Ny = bilerp(nraw + hw, m.w, ty, tx);
    Nz = bilerp(nraw + 2 * hw, m.w, ty, tx);
}

// per full-resolution pixel: G (for the quantile, twice: once to keep, once as sort key) and t = 1 - <n^, Dn>
__global__ void __launch_bounds__(256) dn_fullres_kernel(Dims m, const float *__restrict__ nraw, const float *__restrict__ gnorm,
                                                          const float *__restrict__ normal, float *__restrict__ G, uint32_t *__restrict__ Gkey,
                                                          float *__restrict__ t, SelState *__restrict__ sel)
{
    const int k = blockIdx.x * 256 + threadIdx.x, HW = m.H * m.W;
    uint32_t my_key = 0u;
    const bool live = k < HW;
    if (live)
    {
    const int y = k / m.W, x = k - y * m.W;
    float Nx, Ny, Nz;
    fullres_normal(m, nraw, y, x, Nx, Ny, Nz);
    const float inv = 1.0f / sqrtf(Nx * Nx + Ny * Ny + Nz * Nz);
    const float nx = normal[k], ny = normal[HW + k], nz = normal[2 * HW + k];
    const float nn = fmaxf(sqrtf(nx * nx + ny * ny + nz * nz), 1e-8f); // F.normalize(p = 2, dim = 0, eps = 1e-8)
    const float dot = (nx * Nx + ny * Ny + nz * Nz) * inv / nn;
    float g;
    if (m.h == m.H && m.w == m.W) g = gnorm[k];
    else g = bilerp(gnorm, m.w, tap_of(y, m.r_up_y, m.h), tap_of(x, m.r_up_x, m.w));
    G[k] = g;
    my_key = __float_as_uint(g);
    Gkey[k] = my_key; // G >= 0: the bit pattern is monotone
    t[k] = 1.0f - dot;
    }
    // (the select's first histogram pass was fused in here and taken out again: one key per thread means 256 global adds per 256 keys on the same 256
    //  counters -- sixteen times the standalone kernel's -- and depth_normal_fwd went 0.130 -> 0.150 ms, profiles/r06_loss_bench.jsonl)
    (void)sel; (void)my_key;
}

__global__ void __launch_bounds__(256) dn_sum_kernel(int HW, const float *__restrict__ t, const float *__restrict__ G, float *__restrict__ thr,
                                                      double *__restrict__ partial, const SelState *__restrict__ sel, float q)
{
    __shared__ double red[4];
    // torch.quantile from the finished select state: every block works it out for itself (four 256-thread scans), block 0 leaves it for the backward
    const float th = sel_threshold_value(sel, (size_t)HW, q);
    if (blockIdx.x == 0 && threadIdx.x == 0) *thr = th;
    double s = 0.0;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < HW; k += gridDim.x * 256)
        if (G[k] < th) s += (double)t[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// One wave, fixed order (lane l adds partials l, l + 64, ...; then a butterfly): deterministic from run to run.  Until round 6 ONE thread walked the up
// to 1024 partials -- a chain of dependent global loads that took 51 us, the longest kernel of the whole loss (profiles/r06_depth_normal_kernels.txt).
__global__ void __launch_bounds__(64) dn_finish_kernel(int nblocks, int HW, const double *__restrict__ partial, float *__restrict__ out)
{
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[0] = (float)(s / (double)HW);
}

// ---- backward ----------------------------------------------------------------------------------------------------------------------
// full resolution: dL/dnormal (output) and dL/dN (scratch), L = sum_p m_p (1 - <n^, N / |N|>), m_p = mask_p g / HW
__global__ void __launch_bounds__(256) dn_bwd_fullres_kernel(Dims m, const float *__restrict__ nraw, const float *__restrict__ normal,
                                                              const float *__restrict__ G, const float *__restrict__ thr, const float *__restrict__ grad_out,
                                                              float *__restrict__ dN, float *__restrict__ dnormal)
{
    const int k = blockIdx.x * 256 + threadIdx.x, HW = m.H * m.W;
    if (k >= HW) return;
    const int y = k / m.W, x = k - y * m.W;
    const float go = grad_out ? *grad_out : 1.0f;
    const float w = (G[k] < *thr) ? go / (float)HW : 0.0f;
    float Nx, Ny, Nz;
    fullres_normal(m, nraw, y, x, Nx, Ny, Nz);
    const float inv = 1.0f / sqrtf(Nx * Nx + Ny * Ny + Nz * Nz);
    const float Dx = Nx * inv, Dy = Ny * inv, Dz = Nz * inv;
    const float nx = normal[k], ny = normal[HW + k], nz = normal[2 * HW + k];
    const float len = sqrtf(nx * nx + ny * ny + nz * nz), nn = fmaxf(len, 1e-8f);
    const float hx = nx / nn, hy = ny / nn, hz = nz / nn;
    if (dN)
    {
        // dL/dD^ = -w n^ ;  dL/dN = (dL/dD^ - D^ <D^, dL/dD^>) / |N|
        const float s = -w * (hx * Dx + hy * Dy + hz * Dz);
        dN[k] = (-w * hx - Dx * s) * inv;
        dN[HW + k] = (-w * hy - Dy * s) * inv;
        dN[2 * HW + k] = (-w * hz - Dz * s) * inv;
    }
    if (dnormal)
    {
        // dL/dn^ = -w D^ ;  n^ = n / max(|n|, eps): below eps the denominator is the constant eps
        float gx = -w * Dx, gy = -w * Dy, gz = -w * Dz;
        if (len > 1e-8f)
        {
            const float s = hx * gx + hy * gy + hz * gz;
            gx -= hx * s; gy -= hy * s; gz -= hz * s;
        }
        dnormal[k] = gx / nn;
        dnormal[HW + k] = gy / nn;
        dnormal[2 * HW + k] = gz / nn;
    }
}

// low resolution: adjoint of the up-sampling as a gather, then the chain through (Dx, Dy) to (dgx, dgy, dd_direct)
__global__ void __launch_bounds__(256) dn_bwd_lowres_kernel(Dims m, const float *__restrict__ d, const float *__restrict__ dN, float *__restrict__ dlow)
{
    const int k = blockIdx.x * 256 + threadIdx.x, hw = m.h * m.w, HW = m.H * m.W;
    if (k >= hw) return;
    const int i = k / m.w, j = k - i * m.w;
    float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
    if (m.h == m.H && m.w == m.W) { g0 = dN[k]; g1 = dN[HW + k]; g2 = dN[2 * HW + k]; }
    else
    {
        int ylo, yhi, xlo, xhi;
        dst_range(i, m.r_up_y, m.H, ylo, yhi);
        dst_range(j, m.r_up_x, m.W, xlo, xhi);
        for (int y = ylo; y <= yhi; y++)
        {
            const Tap ty = tap_of(y, m.r_up_y, m.h);
            const float wy = (ty.i0 == i ? ty.l0 : 0.0f) + (ty.i1 == i ? ty.l1 : 0.0f);
            if (wy == 0.0f) continue;
            for (int x = xlo; x <= xhi; x++)
            {
                const Tap tx = tap_of(x, m.r_up_x, m.w);
                const float wx = (tx.i0 == j ? tx.l0 : 0.0f) + (tx.i1 == j ? tx.l1 : 0.0f);
                if (wx == 0.0f) continue;
                const size_t p = (size_t)y * m.W + x;
                const float ww = wy * wx;
                g0 += ww * dN[p]; g1 += ww * dN[HW + p]; g2 += ww * dN[2 * HW + p];
            }
        }
    }
    float gx, gy;
    scharr(d, i, j, m.h, m.w, gx, gy);
    const float dv = d[k], Dx = gx / dv, Dy = gy / dv;
    const float cx = (float)j - 0.5f * (float)m.w + 0.5f, cy = (float)i - 0.5f * (float)m.h + 0.5f;
    const float dDx = m.A * g0 - cx * g2, dDy = m.B * g1 - cy * g2;
    dlow[k] = dDx / dv;                          // dL/dgx
    dlow[hw + k] = dDy / dv;                     // dL/dgy
    dlow[2 * hw + k] = -(dDx * Dx + dDy * Dy) / dv; // dL/dd through the two divisions
}

// Scharr adjoint (zero padding is self-adjoint): dd(r, c) = direct + sum_ab Kx[a][b] dgx(r - a + 1, c - b + 1) + Ky[a][b] dgy(...)
__global__ void __launch_bounds__(256) dn_bwd_scharr_kernel(Dims m, const float *__restrict__ dlow, float *__restrict__ dd)
{
    const int k = blockIdx.x * 256 + threadIdx.x, hw = m.h * m.w;
    if (k >= hw) return;
    const int r = k / m.w, c = k - r * m.w;
    const float *gx = dlow, *gy = dlow + hw;
    // output (r', c') = (r - a + 1, c - b + 1) used input (r, c) with weight K[a][b]: a = r - r' + 1, b = c - c' + 1
    float s = dlow[2 * hw + k];
#pragma unroll
    for (int dr = -1; dr <= 1; dr++)
#pragma unroll
        for (int dc = -1; dc <= 1; dc++)
        {
            const int a = 1 - dr, b = 1 - dc; // kernel element that output (r + dr, c + dc) applied to input (r, c)
            const float kx = (b == 1 ? 0.0f : (b == 0 ? -1.0f : 1.0f)) * (a == 1 ? 10.0f : 3.0f);
            const float ky = (a == 1 ? 0.0f : (a == 0 ? -1.0f : 1.0f)) * (b == 1 ? 10.0f : 3.0f);
            s += (kx * at0(gx, r + dr, c + dc, m.h, m.w) + ky * at0(gy, r + dr, c + dc, m.h, m.w)) * (1.0f / 32.0f);
        }
    dd[k] = s;
}

// full resolution: adjoint of the down-sampling as a gather
__global__ void __launch_bounds__(256) dn_bwd_upsample_kernel(Dims m, const float *__restrict__ dd, float *__restrict__ ddepth)
{
    const int k = blockIdx.x * 256 + threadIdx.x, HW = m.H * m.W;
    if (k >= HW) return;
    if (m.h == m.H && m.w == m.W) { ddepth[k] = dd[k]; return; }
    const int y = k / m.W, x = k - y * m.W;
    int ilo, ihi, jlo, jhi;
    dst_range(y, m.r_down, m.h, ilo, ihi);
    dst_range(x, m.r_down, m.w, jlo, jhi);
    float s = 0.0f;
    for (int i = ilo; i <= ihi; i++)
    {
        const Tap ty = tap_of(i, m.r_down, m.H);
        const float wy = (ty.i0 == y ? ty.l0 : 0.0f) + (ty.i1 == y ? ty.l1 : 0.0f);
        if (wy == 0.0f) continue;
        for (int j = jlo; j <= jhi; j++)
        {
            const Tap tx = tap_of(j, m.r_down, m.W);
            const float wx = (tx.i0 == x ? tx.l0 : 0.0f) + (tx.i1 == x ? tx.l1 : 0.0f);
            if (wx != 0.0f) s += wy * wx * dd[(size_t)i * m.w + j];
        }
    }
    ddepth[k] = s;
}

struct Carve
{
    float *d, *nraw, *gnorm, *G, *t, *thr, *dN, *dlow, *dd;
    uint32_t *k[2], *v[2];
    double *partial;
    void *scratch;
    size_t bytes;
};
constexpr int SUM_BLOCKS = 1024;
Dims make_dims(int H, int W, float tan_fovx, float tan_fovy, double scale)
{
    Dims m;
    m.H = H; m.W = W;
    // scale_factor travels as a double, like the Python float torch receives: the output size floor(H * s) and the coordinate ratio
    // (float)(1.0 / s) are formed in double exactly as F.interpolate forms them (H = 10, s = 0.7 gives 7 rows; the float 0.69999999 gave 6)
    const bool same = !(scale > 0.0) || scale == 1.0;
    m.h = same ? H : (int)floor((double)H * scale);
    m.w = same ? W : (int)floor((double)W * scale);
    m.r_down = same ? 1.0f : (float)(1.0 / scale);   // interpolate(scale_factor = s): the given factor maps coordinates
    m.r_up_y = (float)m.h / (float)H;        // interpolate(size = ...): the size ratio does
    m.r_up_x = (float)m.w / (float)W;
    m.A = (float)m.w / (2.0f * tan_fovx);
    m.B = (float)m.h / (2.0f * tan_fovy);
    return m;
}
Carve carve(void *ws, const Dims &m)
{
    Carve c;
    char *p = (char *)ws;
    const size_t hw = (size_t)m.h * m.w, HW = (size_t)m.H * m.W;
    ts_carve(p, c.d, hw);
    ts_carve(p, c.nraw, 3 * hw);
    ts_carve(p, c.gnorm, hw);
    ts_carve(p, c.G, HW);
    ts_carve(p, c.t, HW);
    ts_carve(p, c.thr, (size_t)4);
    ts_carve(p, c.partial, (size_t)SUM_BLOCKS);
    ts_carve(p, c.dN, 3 * HW);
    ts_carve(p, c.dlow, 3 * hw);
    ts_carve(p, c.dd, hw);
    for (int i = 0; i < 2; i++) { ts_carve(p, c.k[i], HW); ts_carve(p, c.v[i], HW); }
    p = (char *)ts_align_up((size_t)p);
    c.scratch = p;
    p += ts_radix_scratch_bytes(HW) > ts_quantile_scratch_bytes() ? ts_radix_scratch_bytes(HW) : ts_quantile_scratch_bytes(); // (the quantile is a radix select since round 6; the sort scratch is kept as an upper bound)
    c.bytes = (size_t)(p - (char *)ws) + TS_ALIGN;
    return c;
}
} // namespace

size_t ts_depth_normal_workspace_bytes(int H, int W, double scale)
{
    if (H <= 0 || W <= 0) return TS_ALIGN;
    return carve(nullptr, make_dims(H, W, 1.0f, 1.0f, scale)).bytes;
}

hipError_t ts_depth_normal_forward(const float *depth, const float *normal, int H, int W, float tan_fovx, float tan_fovy, double scale, float quantile,
                                   void *workspace, float *out, hipStream_t s)
{
    const Dims m = make_dims(H, W, tan_fovx, tan_fovy, scale);
    const Carve c = carve(workspace, m);
    const int hw = m.h * m.w, HW = H * W;
    const dim3 lo((unsigned)((hw + 255) / 256)), hi((unsigned)((HW + 255) / 256));
    // torch.quantile by radix select (select.hip; the norms are >= 0, their bit patterns order like the values), woven into this sequence: the state is
    // cleared by the first kernel and the kernel that needs the threshold forms it (13 -> 11 dependent launches; at 800 x 800 every launch is ~8 us of
    // latency, not work)
    SelState *sel = (SelState *)ts_align_up((size_t)c.scratch);
    const int nsel = (int)ts_quantile_state_words();
    const dim3 lo0((unsigned)((max(hw, nsel) + 255) / 256));
    hipLaunchKernelGGL(dn_downsample_kernel, lo0, dim3(256), 0, s, m, depth, c.d, (uint32_t *)sel, nsel);
    hipLaunchKernelGGL(dn_lowres_kernel, lo, dim3(256), 0, s, m, c.d, c.nraw, c.gnorm);
    hipLaunchKernelGGL(dn_fullres_kernel, hi, dim3(256), 0, s, m, c.nraw, c.gnorm, normal, c.G, c.k[0], c.t, sel);
    ts_quantile_passes(c.k[0], (size_t)HW, quantile, c.scratch, 0, s);
    const int nb = min(SUM_BLOCKS, (HW + 255) / 256);
    hipLaunchKernelGGL(dn_sum_kernel, dim3((unsigned)nb), dim3(256), 0, s, HW, c.t, c.G, c.thr, c.partial, sel, quantile);
    hipLaunchKernelGGL(dn_finish_kernel, dim3(1), dim3(64), 0, s, nb, HW, c.partial, out);
    return hipGetLastError();
}

hipError_t ts_depth_normal_backward(const float *depth, const float *normal, int H, int W, float tan_fovx, float tan_fovy, double scale,
                                    const void *workspace, const float *grad_out, float *dL_ddepth, float *dL_dnormal, hipStream_t s)
{
    (void)depth;
    const Dims m = make_dims(H, W, tan_fovx, tan_fovy, scale);
    const Carve c = carve(const_cast<void *>(workspace), m);
    const int hw = m.h * m.w, HW = H * W;
    const dim3 lo((unsigned)((hw + 255) / 256)), hi((unsigned)((HW + 255) / 256));
    hipLaunchKernelGGL(dn_bwd_fullres_kernel, hi, dim3(256), 0, s, m, c.nraw, normal, c.G, c.thr, grad_out, dL_ddepth ? c.dN : nullptr, dL_dnormal);
    if (dL_ddepth)
    {
        hipLaunchKernelGGL(dn_bwd_lowres_kernel, lo, dim3(256), 0, s, m, c.d, c.dN, c.dlow);
        hipLaunchKernelGGL(dn_bwd_scharr_kernel, lo, dim3(256), 0, s, m, c.dlow, c.dd);
        hipLaunchKernelGGL(dn_bwd_upsample_kernel, hi, dim3(256), 0, s, m, c.dd, dL_ddepth);
    }
    return hipGetLastError();
}
