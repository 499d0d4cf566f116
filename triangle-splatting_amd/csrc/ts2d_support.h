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
// pad_px: an extra acceptance margin in pixels on the three edge tests (0 for the 2D variant, whose per-pixel test IS the fp32 evaluation of
// these barycentrics; the 3D variant's per-pixel test goes through the ray / plane intersection, whose rounding it has to cover)
__device__ __forceinline__ QuadSetup quad_setup(float v1x, float v1y, float v2x, float v2y, float v3x, float v3y, float E, float pad_px = 0.0f)
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
    const float slope_margin = 2e-6f * 15.0f + pad_px;
    q.P1 = fmaxf(0.0f, 7.0f * q.A1) + fmaxf(0.0f, 7.0f * q.B1) - m + slope_margin * (fabsf(q.A1) + fabsf(q.B1));
    q.P2 = fmaxf(0.0f, 7.0f * q.A2) + fmaxf(0.0f, 7.0f * q.B2) - m + slope_margin * (fabsf(q.A2) + fabsf(q.B2));
    q.P3 = fmaxf(0.0f, 7.0f * q.A3) + fmaxf(0.0f, 7.0f * q.B3) - m + slope_margin * (fabsf(q.A3) + fabsf(q.B3));
    const float cx = (v1x + v2x + v3x) * (1.0f / 3.0f), cy = (v1y + v2y + v3y) * (1.0f / 3.0f);
    const float e1x = E * (v1x - cx), e2x = E * (v2x - cx), e3x = E * (v3x - cx);
    const float e1y = E * (v1y - cy), e2y = E * (v2y - cy), e3y = E * (v3y - cy);
    // 0.05 px like the block cull + the rounding of absolute coordinates (|c| ulp each for c and c + e)
    const float padx = 0.05f + pad_px + 4e-7f * fabsf(cx), pady = 0.05f + pad_px + 4e-7f * fabsf(cy);
    q.bminx = cx + fminf(fminf(e1x, e2x), e3x) - padx; q.bmaxx = cx + fmaxf(fmaxf(e1x, e2x), e3x) + padx;
    q.bminy = cy + fminf(fminf(e1y, e2y), e3y) - pady; q.bmaxy = cy + fmaxf(fmaxf(e1y, e2y), e3y) + pady;
    q.live = E > 0.0f;
    return q;
}
// A setup whose masks are always 0xF: for triangles the test cannot be trusted on (3D: a scaled vertex behind the camera, a projection of no area).
__device__ __forceinline__ QuadSetup quad_setup_all()
{
    QuadSetup q{};
    q.P1 = q.P2 = q.P3 = 1e30f;
    q.bminx = q.bminy = -3e38f;
    q.bmaxx = q.bmaxy = 3e38f;
    q.live = true;
    return q;
}
// 3D variant: the record's view-space triangle scaled by E about its centroid, projected to pixels (R3D auxiliary.h:35-43: ndc = x / (z tan),
// pixel = ((ndc + 1) S - 1) / 2), then the 2D setup with E = 1.  PAD3D covers what the ray / plane arithmetic of render3d_group.hip can move
// a decision by (estimated <= 1e-3 px for all but edge-on triangles, which the area test below sends to quad_setup_all).
constexpr float PAD3D = 0.02f;
// One more case goes to quad_setup_all: the HORIZON of the triangle's plane crossing its tile rectangle.  Where p_ray . n passes through zero
// the reference's arithmetic (and render3d_group.hip's, expression for expression) does not fail cleanly: with 1e-8 <= |p_ray . n| the depth is
// ~1e8, the three p_vk = v_k - depth p_ray round to the SAME vector, both cross products vanish, a = (0, 0, 1), ecc = 1 -- a spurious hit with
// alpha = o exp(-1/2) on the horizon line, anywhere in the rectangle (tools/sim/qmask_model.py found it; cull3 keeps such blocks for the same
// reason).  p_ray . n is affine in the pixel, so its extremes over the rectangle [px0, px1] x [py0, py1] sit at the corners: all quadrants
// unless it keeps its sign there with |.| >= 1e-3 of its largest corner value (the slack cull3 uses).  ~2 % of randomly oriented triangles.
__device__ __forceinline__ QuadSetup quad_setup_3d(const float4 &r0, const float4 &r1, const float4 &r2, float E, float tan_fovx, float tan_fovy, int W, int H,
                                                   float inv_W, float inv_H, float px0, float py0, float px1, float py1)
{
    {
        const float nx = r2.y, ny = r2.z, nz = r2.w;
        const float sx = tan_fovx * inv_W, sy = tan_fovy * inv_H; // multiplications by the reciprocals: a sign / magnitude test with a 1e-3 slack
        const float rx0 = sx * (2.0f * px0 - (float)W + 1.0f), rx1 = sx * (2.0f * px1 - (float)W + 1.0f);
        const float ry0 = sy * (2.0f * py0 - (float)H + 1.0f), ry1 = sy * (2.0f * py1 - (float)H + 1.0f);
        const float d00 = rx0 * nx + ry0 * ny + nz, d10 = rx1 * nx + ry0 * ny + nz, d01 = rx0 * nx + ry1 * ny + nz, d11 = rx1 * nx + ry1 * ny + nz;
        const float lo = fminf(fminf(d00, d10), fminf(d01, d11)), hi = fmaxf(fmaxf(d00, d10), fmaxf(d01, d11));
        const float big = fmaxf(fabsf(lo), fabsf(hi));
        if (!(lo > 1e-3f * big || hi < -1e-3f * big)) return quad_setup_all();
    }
    const float v1x = r0.x, v1y = r0.y, v1z = r0.z, v2x = r0.w, v2y = r1.x, v2z = r1.y, v3x = r1.z, v3y = r1.w, v3z = r2.x;
    const float cx = (v1x + v2x + v3x) * (1.0f / 3.0f), cy = (v1y + v2y + v3y) * (1.0f / 3.0f), cz = (v1z + v2z + v3z) * (1.0f / 3.0f);
    const float w1z = cz + E * (v1z - cz), w2z = cz + E * (v2z - cz), w3z = cz + E * (v3z - cz);
    const float zmin = 0.05f * cz;
    if (!(E > 0.0f && cz > 0.0f && w1z >= zmin && w2z >= zmin && w3z >= zmin)) return quad_setup_all();
    const float kx = 0.5f * (float)W * __builtin_amdgcn_rcpf(tan_fovx), ky = 0.5f * (float)H * __builtin_amdgcn_rcpf(tan_fovy);
    const float ox = 0.5f * (float)W - 0.5f, oy = 0.5f * (float)H - 0.5f;
    const float i1 = __builtin_amdgcn_rcpf(w1z), i2 = __builtin_amdgcn_rcpf(w2z), i3 = __builtin_amdgcn_rcpf(w3z); // 1 ulp: far inside PAD3D
    const float s1x = (cx + E * (v1x - cx)) * i1 * kx + ox, s1y = (cy + E * (v1y - cy)) * i1 * ky + oy;
    const float s2x = (cx + E * (v2x - cx)) * i2 * kx + ox, s2y = (cy + E * (v2y - cy)) * i2 * ky + oy;
    const float s3x = (cx + E * (v3x - cx)) * i3 * kx + ox, s3y = (cy + E * (v3y - cy)) * i3 * ky + oy;
    const float area2 = (s2x - s1x) * (s3y - s1y) - (s2y - s1y) * (s3x - s1x);
    const float span = fmaxf(fmaxf(fabsf(s2x - s1x), fabsf(s3x - s1x)), fmaxf(fabsf(s2y - s1y), fabsf(s3y - s1y)));
    if (!(fabsf(area2) > 1e-3f * span) || !(span < 1e7f)) return quad_setup_all(); // thinner than a thousandth of a pixel (or not finite): edge-on
    return quad_setup(s1x, s1y, s2x, s2y, s3x, s3y, 1.0f, PAD3D);
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

// ---- the same test, affine over a triangle's tile rectangle (round 5: one setup per TRIANGLE instead of one per instance) -----------------
// C_k is affine in the tile origin: C_k(TX0 + dx, TY0 + dy) = C_k(TX0, TY0) + A_k dx + B_k dy.  quad_anchor evaluates the constants once at
// the rectangle's first tile; quadrant_mask_affine then needs two FMAs per edge and no gather.  What the affine step adds in rounding goes into
// the acceptance margin, bounded over the whole rectangle [TX0, TX0 + Wpx] x [TY0, TY0 + Hpx] (origins of its tiles):
//   * r_k: rounding of the blend kernels' own products at ANY tile of the rectangle (quadrant_mask adds it per tile): the coordinates relative
//     to a tile origin of the rectangle are at most U = max(|v - T0|, |v - T0 - extent|);
//   * the slopes carry <= 3 ulp (difference, reciprocal, product) and each FMA rounds once at the magnitude of its result:
//     6e-7 (|A_k| Wpx + |B_k| Hpx) + 2.5e-7 |C_k(anchor)| covers both.
// 16 dwords per triangle: K1 K2 K3 A1 | A2 A3 B1 B2 | B3 bminx bmaxx bminy | bmaxy id (minx | miny << 16) spare.  A dead triangle (E <= 0) gets
// an empty bounding box.  Pinned by tools/sim/qmask_model.py (tests/test_qmask_model_cpu.py) against the per-pixel test, like quadrant_mask.
struct QuadAffine { float4 a, b, c, d; };
__device__ __forceinline__ QuadAffine quad_anchor(const QuadSetup &q, uint32_t id, uint32_t minx, uint32_t miny, uint32_t w, uint32_t h)
{
    const float TX = (float)(minx * TS_TILE), TY = (float)(miny * TS_TILE), Wpx = (float)((w - 1u) * TS_TILE), Hpx = (float)((h - 1u) * TS_TILE);
    const float u1x = q.v1x - TX, u1y = q.v1y - TY, u2x = q.v2x - TX, u2y = q.v2y - TY, u3x = q.v3x - TX, u3y = q.v3y - TY;
    const float aia = fabsf(q.ia);
    const float C1 = (u2x * u3y - u2y * u3x) * q.ia, C2 = (u3x * u1y - u3y * u1x) * q.ia;
    const float C3 = 1.0f - C1 - C2;
    const float U1x = fmaxf(fabsf(u1x), fabsf(u1x - Wpx)), U1y = fmaxf(fabsf(u1y), fabsf(u1y - Hpx));
    const float U2x = fmaxf(fabsf(u2x), fabsf(u2x - Wpx)), U2y = fmaxf(fabsf(u2y), fabsf(u2y - Hpx));
    const float U3x = fmaxf(fabsf(u3x), fabsf(u3x - Wpx)), U3y = fmaxf(fabsf(u3y), fabsf(u3y - Hpx));
    const float r1 = 4e-7f * (U2x * U3y + U2y * U3x) * aia, r2 = 4e-7f * (U3x * U1y + U3y * U1x) * aia;
    const float s1 = 6e-7f * (fabsf(q.A1) * Wpx + fabsf(q.B1) * Hpx) + 2.5e-7f * fabsf(C1);
    const float s2 = 6e-7f * (fabsf(q.A2) * Wpx + fabsf(q.B2) * Hpx) + 2.5e-7f * fabsf(C2);
    const float s3 = 6e-7f * (fabsf(q.A3) * Wpx + fabsf(q.B3) * Hpx) + 2.5e-7f * fabsf(C3);
    QuadAffine o;
    o.a = make_float4(C1 + q.P1 + r1 + s1, C2 + q.P2 + r2 + s2, C3 + q.P3 + (r1 + r2 + 4e-7f) + s3, q.A1);
    o.b = make_float4(q.A2, q.A3, q.B1, q.B2);
    o.c = make_float4(q.B3, q.live ? q.bminx : 3e38f, q.live ? q.bmaxx : -3e38f, q.bminy);
    o.d = make_float4(q.bmaxy, __uint_as_float(id), __uint_as_float(minx | (miny << 16)), 0.0f);
    return o;
}
// (dx, dy) = the tile's offset inside the rectangle, in tiles; (x, y) = its absolute tile coordinates
__device__ __forceinline__ uint32_t quadrant_mask_affine(const QuadAffine &o, uint32_t dx, uint32_t dy, uint32_t x, uint32_t y)
{
    const float fx = (float)(dx * TS_TILE), fy = (float)(dy * TS_TILE), TX = (float)(x * TS_TILE), TY = (float)(y * TS_TILE);
    const float A1 = o.a.w, A2 = o.b.x, A3 = o.b.y, B1 = o.b.z, B2 = o.b.w, B3 = o.c.x;
    const float k1 = fmaf(B1, fy, fmaf(A1, fx, o.a.x)), k2 = fmaf(B2, fy, fmaf(A2, fx, o.a.y)), k3 = fmaf(B3, fy, fmaf(A3, fx, o.a.z));
    const float ax1 = 8.0f * A1, ax2 = 8.0f * A2, ax3 = 8.0f * A3, by1 = 8.0f * B1, by2 = 8.0f * B2, by3 = 8.0f * B3;
    const float bminx = o.c.y, bmaxx = o.c.z, bminy = o.c.w, bmaxy = o.d.x;
    const bool x0 = bminx <= TX + 7.0f && bmaxx >= TX, x1 = bminx <= TX + 15.0f && bmaxx >= TX + 8.0f;
    const bool y0 = bminy <= TY + 7.0f && bmaxy >= TY, y1 = bminy <= TY + 15.0f && bmaxy >= TY + 8.0f;
    uint32_t m = 0;
    m |= (x0 && y0 && k1 >= 0.0f && k2 >= 0.0f && k3 >= 0.0f) ? 1u : 0u;
    m |= (x1 && y0 && k1 + ax1 >= 0.0f && k2 + ax2 >= 0.0f && k3 + ax3 >= 0.0f) ? 2u : 0u;
    m |= (x0 && y1 && k1 + by1 >= 0.0f && k2 + by2 >= 0.0f && k3 + by3 >= 0.0f) ? 4u : 0u;
    m |= (x1 && y1 && k1 + ax1 + by1 >= 0.0f && k2 + ax2 + by2 >= 0.0f && k3 + ax3 + by3 >= 0.0f) ? 8u : 0u;
    return m;
}
} // namespace
