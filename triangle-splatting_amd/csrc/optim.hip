// optim.hip -- fused Adam step over up to TSO_MAX_SLICES flat ranges in one launch (include/ts_optim.h).
//
// Replaces torch.optim.Adam.step() as the reference's trainer calls it (src/diff_recon/trainers/VanillaTS_trainer.py:119-122; groups and
// eps: src/diff_recon/models/VanillaTS_model.py:108-124).  HBM-bound: per element 16 bytes in (param, grad, two moments), 12 out.  One
// workgroup = 256 lanes x 4 consecutive floats of ONE slice; slices whose four pointers are 16-byte aligned move dwordx4, the others
// dwords (a slice of a flat buffer may start at any multiple of 4 bytes).  Built with -ffp-contract=off: torch's operation order, every
// operation rounded on its own.
#include "ts2d_common.h"
#include "ts2d_sh.h"
#include "../../include/ts_optim.h"

using namespace ts;

namespace
{
struct AdamTable
{
    tso_adam_slice s[TSO_MAX_SLICES];
    unsigned first_block[TSO_MAX_SLICES + 1]; // slice k owns blocks [first_block[k], first_block[k + 1])
    unsigned char vec4[TSO_MAX_SLICES];
    int n;
    float beta1, beta2, w1, w2, eps;
};

struct AdamCoeffs
{
    float beta2, w1, w2, eps;
};

__device__ __forceinline__ void adam_element(float &p, float g, float &m, float &v, const AdamCoeffs &t, float step_size, float bias2_sqrt, float grad_scale)
{
    g = g * grad_scale;
    // exp_avg.lerp_(grad, 1 - beta1): ATen's lerp (aten/src/ATen/native/Lerp.h) takes  self + w (end - self)  for |w| < 0.5 and
    // end - (end - self)(1 - w)  otherwise (beta1 <= 0.5) -- both forms, so that the 1-ulp statement of include/ts_optim.h holds for every beta1
    m = t.w1 < 0.5f ? m + (g - m) * t.w1 : g - (g - m) * (1.0f - t.w1);
    v = v * t.beta2;                              // exp_avg_sq.mul_(beta2)
    v = v + (t.w2 * g) * g;                       // .addcmul_(grad, grad, value = 1 - beta2)
    const float denom = __fsqrt_rn(v) / bias2_sqrt + t.eps;
    p = p - step_size * (m / denom);              // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ void __launch_bounds__(256) adam_kernel(const AdamTable t)
{
    int k = 0;
#pragma unroll
    for (int i = 1; i < TSO_MAX_SLICES; i++)
        if (i < t.n && blockIdx.x >= t.first_block[i]) k = i;
    const tso_adam_slice &s = t.s[k];
    const int64_t i0 = ((int64_t)(blockIdx.x - t.first_block[k]) * 256 + threadIdx.x) * 4;
    if (i0 >= s.count) return;
    const int n = (int)(s.count - i0 < 4 ? s.count - i0 : 4);
    float p[4], g[4], m[4], v[4];
    if (t.vec4[k] && n == 4)
    {
        const float4 P4 = *(const float4 *)(s.param + i0), G4 = *(const float4 *)(s.grad + i0), M4 = *(const float4 *)(s.exp_avg + i0),
                     V4 = *(const float4 *)(s.exp_avg_sq + i0);
        p[0] = P4.x; p[1] = P4.y; p[2] = P4.z; p[3] = P4.w;
        g[0] = G4.x; g[1] = G4.y; g[2] = G4.z; g[3] = G4.w;
        m[0] = M4.x; m[1] = M4.y; m[2] = M4.z; m[3] = M4.w;
        v[0] = V4.x; v[1] = V4.y; v[2] = V4.z; v[3] = V4.w;
    }
    else
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < n) { p[j] = s.param[i0 + j]; g[j] = s.grad[i0 + j]; m[j] = s.exp_avg[i0 + j]; v[j] = s.exp_avg_sq[i0 + j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (j < n)
        {
            float step = s.step_size;
            if (s.period > 0 && (int)((s.index0 + i0 + j) % s.period) >= s.split) step = s.step_size_tail;
            adam_element(p[j], g[j], m[j], v[j], AdamCoeffs{t.beta2, t.w1, t.w2, t.eps}, step, s.bias2_sqrt, s.grad_scale);
        }
    if (t.vec4[k] && n == 4)
    {
        *(float4 *)(s.param + i0) = make_float4(p[0], p[1], p[2], p[3]);
        *(float4 *)(s.exp_avg + i0) = make_float4(m[0], m[1], m[2], m[3]);
        *(float4 *)(s.exp_avg_sq + i0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    else
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (j < n) { s.param[i0 + j] = p[j]; s.exp_avg[i0 + j] = m[j]; s.exp_avg_sq[i0 + j] = v[j]; }
    }
}

// ---- Adam on the SH coefficients from their FACTORED gradient (include/ts_optim.h: tso_adam_step_sh_factored) -------------------------------
// dL_dshs[i, k, :] = sum_v basis_k(normalize(centre_i - campos_v)) * dL_dRGB_v[i, :] (backward.cu:9-119; shgrad.hip writes exactly this out as a
// dense (P, M, 3) array).  Here the product is formed in registers by the thread that owns coefficient k of triangle i and fed to adam_element:
// the 12 M bytes per triangle of the dense gradient are neither written by the backward nor read here (1 M triangles, degree 3: 192 MB less
// written, 144 MB less read per iteration).  Same expressions in the same order as sh_grad_expand_kernel / sh_grad_store, same -ffp-contract=off:
// the gradient value is bit-identical to the dense one, hence the updated parameters and moments are too.
struct ShFactoredArgs
{
    tso_sh_factored_step a;
    float beta2, w1, w2, eps;
};

template <int MAXDEG>
__global__ void __launch_bounds__(256) adam_sh_factored_kernel(const ShFactoredArgs t)
{
    constexpr int M = (MAXDEG + 1) * (MAXDEG + 1);
    const tso_sh_factored_step &a = t.a;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = q / M;
    const int k = (int)(q - i * M);
    if (i >= a.P) return;
    f3 g = {0.0f, 0.0f, 0.0f};
    const int NB = (a.sh_degree + 1) * (a.sh_degree + 1);
    if (k < NB)
    {
        const float *vp = a.vertex + 9 * i;
        const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]};
        const f3 center = divf(add(add(v1, v2), v3), 3.0f); // forward.cu:87
        for (int v = 0; v < a.V; v++)
        {
            const f3 cp = {a.campos[3 * v], a.campos[3 * v + 1], a.campos[3 * v + 2]};
            const float *gp = a.dL_dcolor + ((size_t)v * a.P + i) * 3;
            const f3 c = {gp[0], gp[1], gp[2]};
            const f3 dir_orig = sub(center, cp);
            const f3 dir = divf(dir_orig, norm(dir_orig));
            float b[16] = {};
            sh_basis(MAXDEG < a.sh_degree ? MAXDEG : a.sh_degree, dir, b); // the first bound: lets the compiler drop the degrees this M cannot hold
            float bk = b[0];
#pragma unroll
            for (int j = 1; j < M; j++) bk = (k == j) ? b[j] : bk;
            g = add(g, scale(bk, c));
        }
    }
    const bool rest = k > 0;
    const int64_t off = rest ? i * a.rest_stride + 3 * (k - 1) : i * a.dc_stride;
    float *P_ = (rest ? a.param_rest : a.param_dc) + off, *M_ = (rest ? a.exp_avg_rest : a.exp_avg_dc) + off,
          *V_ = (rest ? a.exp_avg_sq_rest : a.exp_avg_sq_dc) + off;
    const float step = rest ? a.step_size_rest : a.step_size_dc, b2 = rest ? a.bias2_sqrt_rest : a.bias2_sqrt_dc;
    AdamCoeffs co{t.beta2, t.w1, t.w2, t.eps};
    float p[3] = {P_[0], P_[1], P_[2]}, m[3] = {M_[0], M_[1], M_[2]}, vv[3] = {V_[0], V_[1], V_[2]};
    const float gg[3] = {g.x, g.y, g.z};
#pragma unroll
    for (int c = 0; c < 3; c++) adam_element(p[c], gg[c], m[c], vv[c], co, step, b2, a.grad_scale);
#pragma unroll
    for (int c = 0; c < 3; c++) { P_[c] = p[c]; M_[c] = m[c]; V_[c] = vv[c]; }
}

// The same step for ONE (P, M, 3) tensor with 3 M a multiple of four floats (M = 4, 16) on 16-byte aligned rows: dwordx4 loads and stores (the
// per-coefficient kernel above moves single dwords 12 bytes apart: 5.2 TB/s at 1 M triangles against 6.1 of the dense kernel).  A workgroup owns 64
// triangles: wave w forms coefficients 4 w .. 4 w + 3 of their gradient rows -- one lane per triangle: direction, basis, 12 products -- into LDS, then all four waves
// walk the 64 x 3 M / 4 float4 of the three arrays.  (A first version let each of a row's 12 float4 threads form the basis for itself: twelve times instead of four times
// the arithmetic, hidden under the memory time but not free -- the blend kernels of the next iteration ran 5-8 % slower on a power-limited box,
// profiles/r06_train_step.jsonl.)  Rows of 3 M + 4 floats in LDS: 16-byte chunks 13 i + c, distinct over eight neighbouring lanes.
template <int MAXDEG>
__global__ void __launch_bounds__(256) adam_sh_factored_vec4_kernel(const ShFactoredArgs t)
{
    constexpr int M = (MAXDEG + 1) * (MAXDEG + 1), Q = 3 * M / 4, LROW = 3 * M + 4; // Q float4 per triangle
    static_assert((3 * M) % 4 == 0, "rows of 3 M floats must be whole float4");
    __shared__ __attribute__((aligned(16))) float grow[64 * LROW];
    const tso_sh_factored_step &a = t.a;
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int wave = threadIdx.x >> 6; // wave w forms coefficients 4 w .. 4 w + 3 of the 64 rows (M = 4: wave 0 alone)
    if (4 * wave < M)
    {
        const int64_t i = i0 + (threadIdx.x & 63);
        f3 acc[4] = {{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}};
        if (i < a.P && 4 * wave < (a.sh_degree + 1) * (a.sh_degree + 1))
        {
            const float *vp = a.vertex + 9 * i;
            const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]};
            const f3 center = divf(add(add(v1, v2), v3), 3.0f); // forward.cu:87
            for (int v = 0; v < a.V; v++)
            {
                const f3 cp = {a.campos[3 * v], a.campos[3 * v + 1], a.campos[3 * v + 2]};
                const float *gp = a.dL_dcolor + ((size_t)v * a.P + i) * 3;
                const f3 c = {gp[0], gp[1], gp[2]};
                const f3 dir_orig = sub(center, cp);
                const f3 dir = divf(dir_orig, norm(dir_orig));
                float b[16] = {};
                const int nb = sh_basis(MAXDEG < a.sh_degree ? MAXDEG : a.sh_degree, dir, b);
#pragma unroll
                for (int u = 0; u < 4; u++)
                {
                    float bk = b[u]; // wave-uniform choice of the quadruple
#pragma unroll
                    for (int w = 1; w < M / 4; w++) bk = (wave == w) ? b[4 * w + u] : bk;
                    if (4 * wave + u < nb) acc[u] = add(acc[u], scale(bk, c)); // coefficients above the active degree: zeros (the dense array's)
                }
            }
        }
        float *row = grow + (threadIdx.x & 63) * LROW + 12 * wave; // four coefficients = three float4
        *(float4 *)(row) = make_float4(acc[0].x, acc[0].y, acc[0].z, acc[1].x);
        *(float4 *)(row + 4) = make_float4(acc[1].y, acc[1].z, acc[2].x, acc[2].y);
        *(float4 *)(row + 8) = make_float4(acc[2].z, acc[3].x, acc[3].y, acc[3].z);
    }
    __syncthreads();
    const AdamCoeffs co{t.beta2, t.w1, t.w2, t.eps};
    const int64_t left = a.P - i0;
    const int n4 = (int)(left < 64 ? left : 64) * Q;
#pragma unroll
    for (int u = 0; u < (64 * Q + 255) / 256; u++)
    {
        const int q = threadIdx.x + 256 * u; // float4 q of this workgroup's contiguous 64 x 3 M floats
        if (q >= n4) break;
        const int il = q / Q, j = q - il * Q;
        const float4 G4 = *(const float4 *)(grow + il * LROW + 4 * j);
        const int64_t off = i0 * (3 * M) + 4 * (int64_t)q;
        float4 *P_ = (float4 *)(a.param_dc + off), *M_ = (float4 *)(a.exp_avg_dc + off), *V_ = (float4 *)(a.exp_avg_sq_dc + off);
        const float4 P4 = *P_, M4 = *M_, V4 = *V_;
        float p[4] = {P4.x, P4.y, P4.z, P4.w}, m[4] = {M4.x, M4.y, M4.z, M4.w}, vv[4] = {V4.x, V4.y, V4.z, V4.w};
        const float g[4] = {G4.x, G4.y, G4.z, G4.w};
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const bool dc = j == 0 && e < 3;
            adam_element(p[e], g[e], m[e], vv[e], co, dc ? a.step_size_dc : a.step_size_rest, dc ? a.bias2_sqrt_dc : a.bias2_sqrt_rest, a.grad_scale);
        }
        *P_ = make_float4(p[0], p[1], p[2], p[3]);
        *M_ = make_float4(m[0], m[1], m[2], m[3]);
        *V_ = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    // the other per-triangle parameters of these 64 triangles, from their dense gradients (the barrier above lies between this workgroup's reads of
    // its vertices and these writes; no other workgroup looks at them)
    for (int r = 0; r < a.num_rows; r++)
    {
        const tso_row_slice &rs = a.rows[r];
        const int64_t f0 = i0 * rs.floats_per_row;
        const int nf = (int)(left < 64 ? left : 64) * rs.floats_per_row;
        const bool vec = (((size_t)rs.param | (size_t)rs.grad | (size_t)rs.exp_avg | (size_t)rs.exp_avg_sq) & 15) == 0 && ((f0 & 3) == 0);
        for (int e0 = 4 * (int)threadIdx.x; e0 < nf; e0 += 4 * 256)
        {
            const int cnt = nf - e0 < 4 ? nf - e0 : 4;
            float p[4], g[4], m[4], vv[4];
            if (vec && cnt == 4)
            {
                const float4 P4 = *(const float4 *)(rs.param + f0 + e0), G4 = *(const float4 *)(rs.grad + f0 + e0), M4 = *(const float4 *)(rs.exp_avg + f0 + e0),
                             V4 = *(const float4 *)(rs.exp_avg_sq + f0 + e0);
                p[0] = P4.x; p[1] = P4.y; p[2] = P4.z; p[3] = P4.w; g[0] = G4.x; g[1] = G4.y; g[2] = G4.z; g[3] = G4.w;
                m[0] = M4.x; m[1] = M4.y; m[2] = M4.z; m[3] = M4.w; vv[0] = V4.x; vv[1] = V4.y; vv[2] = V4.z; vv[3] = V4.w;
            }
            else
            {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j < cnt) { p[j] = rs.param[f0 + e0 + j]; g[j] = rs.grad[f0 + e0 + j]; m[j] = rs.exp_avg[f0 + e0 + j]; vv[j] = rs.exp_avg_sq[f0 + e0 + j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (j < cnt) adam_element(p[j], g[j], m[j], vv[j], co, rs.step_size, rs.bias2_sqrt, rs.grad_scale);
            if (vec && cnt == 4)
            {
                *(float4 *)(rs.param + f0 + e0) = make_float4(p[0], p[1], p[2], p[3]);
                *(float4 *)(rs.exp_avg + f0 + e0) = make_float4(m[0], m[1], m[2], m[3]);
                *(float4 *)(rs.exp_avg_sq + f0 + e0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            }
            else
            {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j < cnt) { rs.param[f0 + e0 + j] = p[j]; rs.exp_avg[f0 + e0 + j] = m[j]; rs.exp_avg_sq[f0 + e0 + j] = vv[j]; }
            }
        }
    }
}
} // namespace

hipError_t ts_optim_adam_step_sh_factored(const tso_sh_factored_step &a, double beta1, double beta2, double eps, hipStream_t s)
{
    if (a.P <= 0) return hipSuccess;
    ShFactoredArgs t{a, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps};
    // one tensor (coefficient 0 in front of the others in rows of 3 M floats), rows of whole float4, 16-byte aligned: the dwordx4 kernel
    const bool one_tensor = a.M > 1 && a.dc_stride == 3 * a.M && a.rest_stride == 3 * a.M && a.param_rest == a.param_dc + 3 &&
                            a.exp_avg_rest == a.exp_avg_dc + 3 && a.exp_avg_sq_rest == a.exp_avg_sq_dc + 3;
    if (one_tensor && (a.M == 4 || a.M == 16) && (((size_t)a.param_dc | (size_t)a.exp_avg_dc | (size_t)a.exp_avg_sq_dc) & 15) == 0)
    {
        const dim3 grid4((unsigned)((a.P + 63) / 64)); // 64 triangles per workgroup
        if (a.M == 4) hipLaunchKernelGGL(adam_sh_factored_vec4_kernel<1>, grid4, dim3(256), 0, s, t);
        else hipLaunchKernelGGL(adam_sh_factored_vec4_kernel<3>, grid4, dim3(256), 0, s, t);
        return hipGetLastError();
    }
    t.a.num_rows = 0; // the per-coefficient kernel below does not take them: a launch of the dense kernel behind it (same results)
    const int64_t threads = (int64_t)a.P * a.M;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    switch (a.M)
    {
    case 1: hipLaunchKernelGGL(adam_sh_factored_kernel<0>, grid, block, 0, s, t); break;
    case 4: hipLaunchKernelGGL(adam_sh_factored_kernel<1>, grid, block, 0, s, t); break;
    case 9: hipLaunchKernelGGL(adam_sh_factored_kernel<2>, grid, block, 0, s, t); break;
    case 16: hipLaunchKernelGGL(adam_sh_factored_kernel<3>, grid, block, 0, s, t); break;
    default: return hipErrorInvalidValue;
    }
    if (hipError_t e = hipGetLastError()) return e;
    if (a.num_rows > 0)
    {
        tso_adam_slice sl[TSO_SH_ROW_SLICES] = {};
        for (int r = 0; r < a.num_rows; r++)
        {
            sl[r].param = a.rows[r].param; sl[r].grad = a.rows[r].grad; sl[r].exp_avg = a.rows[r].exp_avg; sl[r].exp_avg_sq = a.rows[r].exp_avg_sq;
            sl[r].count = (int64_t)a.P * a.rows[r].floats_per_row;
            sl[r].step_size = a.rows[r].step_size; sl[r].bias2_sqrt = a.rows[r].bias2_sqrt; sl[r].grad_scale = a.rows[r].grad_scale;
        }
        return ts_optim_adam_step(sl, a.num_rows, beta1, beta2, eps, s);
    }
    return hipSuccess;
}

hipError_t ts_optim_adam_step(const tso_adam_slice *slices, int n, double beta1, double beta2, double eps, hipStream_t s)
{
    AdamTable t{};
    t.n = 0;
    t.beta1 = (float)beta1; t.beta2 = (float)beta2; t.eps = (float)eps;
    t.w1 = (float)(1.0 - beta1); // torch forms 1 - beta as a Python float (double); its kernels receive that rounded to fp32
    t.w2 = (float)(1.0 - beta2);
    unsigned blocks = 0;
    for (int i = 0; i < n; i++)
    {
        if (slices[i].count <= 0) continue;
        const int k = t.n++;
        t.s[k] = slices[i];
        t.first_block[k] = blocks;
        blocks += (unsigned)((slices[i].count + 1023) / 1024);
        const size_t a = (size_t)slices[i].param | (size_t)slices[i].grad | (size_t)slices[i].exp_avg | (size_t)slices[i].exp_avg_sq;
        t.vec4[k] = (a & 15) == 0;
    }
    t.first_block[t.n] = blocks;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, t);
    return hipGetLastError();
}
