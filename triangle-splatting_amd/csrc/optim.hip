// optim.hip -- fused Adam step over up to TSO_MAX_SLICES flat ranges in one launch (include/ts_optim.h).
//
// Replaces torch.optim.Adam.step() as the reference's trainer calls it (src/diff_recon/trainers/VanillaTS_trainer.py:119-122; groups and
// eps: src/diff_recon/models/VanillaTS_model.py:108-124).  HBM-bound: per element 16 bytes in (param, grad, two moments), 12 out.  One
// workgroup = 256 lanes x 4 consecutive floats of ONE slice; slices whose four pointers are 16-byte aligned move dwordx4, the others
// dwords (a slice of a flat buffer may start at any multiple of 4 bytes).  Built with -ffp-contract=off: torch's operation order, every
// operation rounded on its own.
#include "ts2d_common.h"
#include "../../include/ts_optim.h"

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

__device__ __forceinline__ void adam_element(float &p, float g, float &m, float &v, const AdamTable &t, float step_size, float bias2_sqrt, float grad_scale)
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
            adam_element(p[j], g[j], m[j], v[j], t, step, s.bias2_sqrt, s.grad_scale);
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
} // namespace

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
