// model_update.hip -- fused per-iteration densification statistics (include/ts_model.h; VanillaTS_model.py:347-363).
// One lane per triangle, all views of the step folded in registers, every state array read and written once:
// 6 floats in + 6 out + 5 per view per triangle.
#include "../../include/ts_model.h"
#include "ts2d_common.h"

namespace
{
__global__ void __launch_bounds__(256) training_statistic_kernel(int P, int V, const int32_t *__restrict__ radii,
                                                                  const float2 *__restrict__ c2d_grad,
                                                                  const float *__restrict__ csum, const float *__restrict__ cmax,
                                                                  float *__restrict__ g_accum, float *__restrict__ g_denom,
                                                                  float *__restrict__ max_radii, float *__restrict__ s_csum,
                                                                  float *__restrict__ s_cmax, float *__restrict__ c_denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float ga = 0.0f, cnt = 0.0f, mr = -1.0f, ms = 0.0f, mm = 0.0f;
    for (int v = 0; v < V; v++)
    {
        const size_t o = (size_t)v * P + i;
        const int r = radii[o];
        if (r <= 0) continue; // visible_mask = radii > 0 (VanillaTS_model.py:679)
        const float2 g = c2d_grad[o];
        ga += sqrtf(g.x * g.x + g.y * g.y); // :358
        cnt += 1.0f;                        // :359, :362
        mr = fmaxf(mr, (float)r);           // :363
        if (csum) { ms = fmaxf(ms, csum[o]); mm = fmaxf(mm, cmax[o]); } // :360-361 (contributions are >= 0)
    }
    if (cnt == 0.0f) return; // invisible in every view: state untouched, like the masked assignments
    g_accum[i] += ga;
    g_denom[i] += cnt;
    c_denom[i] += cnt;
    max_radii[i] = fmaxf(max_radii[i], mr);
    if (csum) { s_csum[i] = fmaxf(s_csum[i], ms); s_cmax[i] = fmaxf(s_cmax[i], mm); }
}
} // namespace

hipError_t ts_model_training_statistic(int P, int V, const int32_t *radii, const float *c2d_grad, const float *csum, const float *cmax,
                                       float *g_accum, float *g_denom, float *max_radii, float *s_csum, float *s_cmax, float *c_denom,
                                       hipStream_t s)
{
    if (P <= 0 || V <= 0) return hipSuccess;
    hipLaunchKernelGGL(training_statistic_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, V, radii, (const float2 *)c2d_grad, csum,
                       cmax, g_accum, g_denom, max_radii, s_csum, s_cmax, c_denom);
    return hipGetLastError();
}
