// shgrad.hip -- rebuilds dense SH-coefficient gradients from factored per-view colour gradients (multi-GPU exchange).
//
// In the reference every view's dL/dshs is a rank-1 product per triangle: dL_dsh[k] = basis_k(dir) * dL_dRGB with
// dir = normalize(centre - campos) and dL_dRGB the clamp-masked colour gradient (R2D/src/backward.cu:9-119,
// R3D/src/backward.cu:9-118).  Summing V views therefore needs only V * 3 floats per triangle on the wire instead of
// 3 M (48 at degree 3): ranks all-gather (dL_dRGB, campos) and each rebuilds
//     dL_dshs[i, k, :] = sum_v basis_k(normalize(centre_i - campos_v)) * dL_dRGB_v[i, :]
// locally.  For V = 1 the result is bit-identical to what preprocess_bwd writes (same expressions, this file is built
// with -ffp-contract=off like preprocess.hip); for V > 1 it equals the all-reduced dense gradients up to fp32
// summation order.  HBM-bound: reads 36 + 12 V bytes, writes 12 M bytes per triangle.
#include "ts2d_common.h"
#include "ts2d_sh.h"

using namespace ts;

namespace
{
template <int DEG>
__global__ void __launch_bounds__(256) sh_grad_expand_kernel(int P, int M, int V, const float *__restrict__ vertex,
                                                              const float *__restrict__ campos,
                                                              const float *__restrict__ dL_dcolor, float *__restrict__ dL_dshs)
{
    constexpr int NB = (DEG + 1) * (DEG + 1);
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P) return;
    const float *vp = vertex + 9 * (size_t)idx;
    const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]};
    const f3 center = divf(add(add(v1, v2), v3), 3.0f); // forward.cu:87 (both variants use the world-space centroid)
    f3 acc[NB];
#pragma unroll
    for (int k = 0; k < NB; k++) acc[k] = {0.0f, 0.0f, 0.0f};
    for (int v = 0; v < V; v++)
    {
        const f3 cp = {campos[3 * v], campos[3 * v + 1], campos[3 * v + 2]};
        const float *gp = dL_dcolor + ((size_t)v * P + idx) * 3;
        const f3 g = {gp[0], gp[1], gp[2]};
        const f3 dir_orig = sub(center, cp);
        const f3 dir = divf(dir_orig, norm(dir_orig));
        float b[16];
        sh_basis(DEG, dir, b);
#pragma unroll
        for (int k = 0; k < NB; k++) acc[k] = add(acc[k], scale(b[k], g));
    }
    float *o = dL_dshs + (size_t)idx * M * 3;
#pragma unroll
    for (int k = 0; k < NB; k++) st3(o + 3 * k, acc[k]);
    for (int k = NB * 3; k < M * 3; k++) o[k] = 0.0f;
}
} // namespace

void ts_launch_sh_grad_expand(int P, int D, int M, int V, const float *vertex, const float *campos, const float *dL_dcolor,
                              float *dL_dshs, hipStream_t s)
{
    if (P <= 0) return;
    const dim3 grid((P + 255) / 256), block(256);
    switch (D)
    {
    case 0: hipLaunchKernelGGL(sh_grad_expand_kernel<0>, grid, block, 0, s, P, M, V, vertex, campos, dL_dcolor, dL_dshs); break;
    case 1: hipLaunchKernelGGL(sh_grad_expand_kernel<1>, grid, block, 0, s, P, M, V, vertex, campos, dL_dcolor, dL_dshs); break;
    case 2: hipLaunchKernelGGL(sh_grad_expand_kernel<2>, grid, block, 0, s, P, M, V, vertex, campos, dL_dcolor, dL_dshs); break;
    default: hipLaunchKernelGGL(sh_grad_expand_kernel<3>, grid, block, 0, s, P, M, V, vertex, campos, dL_dcolor, dL_dshs); break;
    }
}
