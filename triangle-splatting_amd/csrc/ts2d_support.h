// ts2d_support.h -- where a 2D triangle's window can reach: the scale of its support (shared by the blend kernels' block cull, render_group.hip,
// and the emission kernel, binning.hip) and the QUADRANT MASK of an instance: which of the four 8x8 quadrants of a 16x16 tile
// the support can reach.  The emission kernel stores the mask in the four spare bits of the instance's value (triangle id < 2^28); a quadrant
// wave of the blend kernels then gathers and culls only the entries whose bit is set (36 % of a tile's list on the headline scene).
#pragma once
#include "ts2d_wave.h"

#include "ts2d_common.h" // TS_ID_BITS, TS_ID_MASK

namespace
{
// alpha = min(0.99, o exp(-ecc^(2 gamma) / 2)) >= 1/255 (forward.cu:311-313) needs ecc^(2 gamma) <= 2 ln(255 o); together with ecc <= 10
// (forward.cu:307) the support is the triangle scaled by E about its centroid.  Returns E with a safety margin, or -1: no pixel can pass.
template <bool GAMMA1>
__device__ __forceinline__ float support_scale(float op, float g2)
{
    const float t = 255.0f * op;
    float E = -1.0f;
    if (t >= 1.0f)
    {
        const float L = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(t);
        if (GAMMA1) E = __builtin_amdgcn_sqrtf(L);
        else E = (g2 < 1e-6f) ? 10.0f : pow_nonneg(L, 1.0f / g2);
        E = fminf(E * 1.0005f + 0.002f, 10.01f);
    }
    return E;
}

// Per-triangle part of the quadrant test: the affine barycentrics' slopes, the acceptance offsets for an 8x8 sample box, the bounding box of
// the scaled triangle in absolute pixels.
struct QuadSetup
{
    float v1x, v1y, v2x, v2y, v3x, v3y, ia;
    float A1, B1, A2, B2, A3, B3;
    float P1, P2, P3; // max(0, 7 A) + max(0, 7 B) - m + the part of the rounding margin that does not depend on the tile
    float bminx, bmaxx, bminy, bmaxy;
    bool live;
};
__device__ __forceinline__ QuadSetup quad_setup(float v1x, float v1y, float v2x, float v2y, float v3x, float v3y, float E)
{
    QuadSetup q;
    q.v1x = v1x; q.v1y = v1y; q.v2x = v2x; q.v2y = v2y; q.v3x = v3x; q.v3y = v3y;
    const float area2 = __fsub_rn(__fmul_rn(v2x - v1x, v3y - v1y), __fmul_rn(v2y - v1y, v3x - v1x)); // as the blend kernels form it
    q.ia = __builtin_amdgcn_rcpf(area2);
    q.A1 = (v2y - v3y) * q.ia; q.B1 = (v3x - v2x) * q.ia;
    q.A2 = (v3y - v1y) * q.ia; q.B2 = (v1x - v3x) * q.ia;
    q.A3 = -q.A1 - q.A2; q.B3 = -q.B1 - q.B2;
    const float m = (1.0f - E) * (1.0f / 3.0f); // ecc <= E  <=>  min_k a_k >= m
    // rounding of the slopes (2 ulp) times sample offsets up to 15; the tile-dependent part is added in quadrant_mask
    q.P1 = fmaxf(0.0f, 7.0f * q.A1) + fmaxf(0.0f, 7.0f * q.B1) - m + 2e-6f * 15.0f * (fabsf(q.A1) + fabsf(q.B1));
    q.P2 = fmaxf(0.0f, 7.0f * q.A2) + fmaxf(0.0f, 7.0f * q.B2) - m + 2e-6f * 15.0f * (fabsf(q.A2) + fabsf(q.B2));
    q.P3 = fmaxf(0.0f, 7.0f * q.A3) + fmaxf(0.0f, 7.0f * q.B3) - m + 2e-6f * 15.0f * (fabsf(q.A3) + fabsf(q.B3));
    const float cx = (v1x + v2x + v3x) * (1.0f / 3.0f), cy = (v1y + v2y + v3y) * (1.0f / 3.0f);
    const float e1x = E * (v1x - cx), e2x = E * (v2x - cx), e3x = E * (v3x - cx);
    const float e1y = E * (v1y - cy), e2y = E * (v2y - cy), e3y = E * (v3y - cy);
    // 0.05 px like the block cull + the rounding of absolute coordinates (|c| ulp each for c and c + e)
    const float padx = 0.05f + 4e-7f * fabsf(cx), pady = 0.05f + 4e-7f * fabsf(cy);
    q.bminx = cx + fminf(fminf(e1x, e2x), e3x) - padx; q.bmaxx = cx + fmaxf(fmaxf(e1x, e2x), e3x) + padx;
    q.bminy = cy + fminf(fminf(e1y, e2y), e3y) - pady; q.bmaxy = cy + fmaxf(fmaxf(e1y, e2y), e3y) + pady;
    q.live = E > 0.0f;
    return q;
}
// Bit (qy << 1 | qx) = the support can reach a pixel of the quadrant at (TX + 8 qx, TY + 8 qy), (TX, TY) = the tile's origin in pixels.
// Conservative with respect to the per-pixel test of the blend kernels, like their block cull: separating axes = the bounding box of the scaled
// triangle and its three edge normals, evaluated for the 8x8 sample box, rounding errors added to the acceptance margin.
__device__ __forceinline__ uint32_t quadrant_mask(const QuadSetup &q, float TX, float TY)
{
    const float u1x = q.v1x - TX, u1y = q.v1y - TY, u2x = q.v2x - TX, u2y = q.v2y - TY, u3x = q.v3x - TX, u3y = q.v3y - TY;
    // the products, kept apart: their rounding (half an ulp each, the difference, the reciprocal: < 4e-7 of their magnitudes) is what a sliver
    // or a triangle far larger than the tile can amplify beyond any multiple of |C|
    const float t1a = u2x * u3y, t1b = u2y * u3x, t2a = u3x * u1y, t2b = u3y * u1x, aia = fabsf(q.ia);
    const float C1 = (t1a - t1b) * q.ia, C2 = (t2a - t2b) * q.ia;
    const float C3 = 1.0f - C1 - C2;
    const float r1 = 4e-7f * (fabsf(t1a) + fabsf(t1b)) * aia, r2 = 4e-7f * (fabsf(t2a) + fabsf(t2b)) * aia;
    const float k1 = C1 + q.P1 + r1, k2 = C2 + q.P2 + r2, k3 = C3 + q.P3 + (r1 + r2 + 4e-7f);
    const float ax1 = 8.0f * q.A1, ax2 = 8.0f * q.A2, ax3 = 8.0f * q.A3, by1 = 8.0f * q.B1, by2 = 8.0f * q.B2, by3 = 8.0f * q.B3;
    const bool x0 = q.live && q.bminx <= TX + 7.0f && q.bmaxx >= TX, x1 = q.live && q.bminx <= TX + 15.0f && q.bmaxx >= TX + 8.0f;
    const bool y0 = q.bminy <= TY + 7.0f && q.bmaxy >= TY, y1 = q.bminy <= TY + 15.0f && q.bmaxy >= TY + 8.0f;
    uint32_t m = 0;
    m |= (x0 && y0 && k1 >= 0.0f && k2 >= 0.0f && k3 >= 0.0f) ? 1u : 0u;
    m |= (x1 && y0 && k1 + ax1 >= 0.0f && k2 + ax2 >= 0.0f && k3 + ax3 >= 0.0f) ? 2u : 0u;
    m |= (x0 && y1 && k1 + by1 >= 0.0f && k2 + by2 >= 0.0f && k3 + by3 >= 0.0f) ? 4u : 0u;
    m |= (x1 && y1 && k1 + ax1 + by1 >= 0.0f && k2 + ax2 + by2 >= 0.0f && k3 + ax3 + by3 >= 0.0f) ? 8u : 0u;
    return m;
}
} // namespace
