// render3d.hip -- blend kernels of the 3D variant (rasterizer_type="3D").
//
// Behaviour follows FORWARD::renderCUDA / BACKWARD::renderCUDA of the reference's
// submodules/diff-triangle-rasterization-3D ("R3D": src/forward.cu:151-306, src/backward.cu:216-454): for every pixel
// a view-space ray p_ray = (tan_fovx * pixToProj(x), tan_fovy * pixToProj(y), 1) is intersected with the plane of the
// triangle, barycentrics are taken in 3D, and the same ecc / alpha / front-to-back blend as in the 2D variant follows.
// Kept quirks: the normal is unnormalised; accum_normal has no background term; the BACKWARD skip test is on
// G = exp(power), not on alpha (R3D backward.cu:351 vs forward.cu:265), so pairs with G >= 1/255 > alpha that the
// forward skipped still receive (tiny) gradients, exactly like the reference.
//
// Structure is the one of render.hip (one wave64 per 8x8 quadrant, 64-entry batches, ballot + LDS-broadcast constants,
// transpose-reduce of the 16 gradient values, coalesced 64-byte atomic flush).  What is specific here:
//   * a_k(pixel) = N_k(q) / Den(q) with N_k and Den AFFINE in the in-quadrant pixel offset q (projective geometry), so the
//     per-pixel test costs 6 FMA + 1 rcp + 2 mul.  Den = p_ray . n; the N_k come from n x (delta x v1) with delta the
//     ray's offset from the ray through v1 (entry_setup3), which keeps their rounding error at the level of the
//     reference's per-pixel difference form (the naively expanded closed form cancels ~1e4 : 1, and fitting N_k
//     through sampled pixels blows up where a sample ray grazes the plane);
//   * the conservative support test uses the sign of Den over the quadrant: a_k >= m  <=>  s * (N_k - m Den) >= 0.
//   * backward: with q = hit point - v1, every per-pair gradient term of the reference is linear in (1, q) with per-pair
//     scalar weights (zw1, zw2 = the arg-min barycentric's gradient split over a1 / a2, dip = dL_ddepth / Den), so the hot
//     loop forms 20 weighted moments per lane (weights, weights * q, s2, dL_dopacity, rgb, normal), reduces them with the
//     transpose-reduce networks, and once per batch the entry's owning lane turns the 20 sums into the 16 values of the
//     64-byte gradient record with a handful of cross products (instead of five cross products per pixel and entry).
//     Measuring q from v1 (not from the camera) keeps the sums free of the ~depth/edge cancellation an origin-based
//     expansion would have.
#include "ts2d_common.h"
#include "ts2d_wave.h"

namespace
{
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vscale(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Entry3
{
    float a1x, a1y, a1c, a2x, a2y, a2c; // N_k(q) = akx*qx + aky*qy + akc, already divided by n.n
    float dx, dy, dc;                   // Den(q) = p_ray(q) . n
    float d0;                           // v1_view . n  (depth = d0 / Den)
    bool overlap;
};

// Coefficients of the affine numerators.  With u2 = v2 - v1, u3 = v3 - v1, n = u2 x u3 and r = hit point - v1 (in the
// plane), r = a2 u2 + a3 u3, so a2 = r . (u3 x n) / n.n and a3 = r . (n x u2) / n.n.  The hit point is
// (d0 / Den) p_ray, hence Den r = d0 p_ray - Den v1 = n x (p_ray x v1).  Writing p_ray = v1 / v1.z + delta (the ray
// through v1 plus a small in-image offset) the first part drops out exactly and
//     Den r = n x (delta(q) x v1),   delta(q) = delta0 + (qx sx, qy sy, 0)
// which is affine in q, has no pole where the ray grazes the plane, and loses no more digits than the reference's own
// p_vk = v_k - p_view (both subtract quantities that differ by ~ edge / depth).
template <bool GAMMA1>
__device__ __forceinline__ Entry3 entry_setup3(V3 v1, V3 v2, V3 v3, V3 n, float op, float g2, V3 ray0, float sx, float sy)
{
    Entry3 e;
    const float inn = 1.0f / vdot(n, n);
    e.d0 = vdot(v1, n);
    e.dc = vdot(ray0, n);
    e.dx = n.x * sx;
    e.dy = n.y * sy;
    const float iz = 1.0f / v1.z;
    const float ddx = ray0.x - v1.x * iz, ddy = ray0.y - v1.y * iz; // delta0 (its z is 0)
    const V3 m0 = vcross(n, V3{ddy * v1.z, -ddx * v1.z, ddx * v1.y - ddy * v1.x});
    const V3 mx = vcross(n, V3{0.0f, -sx * v1.z, sx * v1.y});
    const V3 my = vcross(n, V3{sy * v1.z, 0.0f, -sy * v1.x});
    const V3 G2 = vscale(inn, vcross(vsub(v3, v1), n)), G3 = vscale(inn, vcross(n, vsub(v2, v1)));
    e.a2c = vdot(m0, G2); e.a2x = vdot(mx, G2); e.a2y = vdot(my, G2);
    const float a3c = vdot(m0, G3), a3x = vdot(mx, G3), a3y = vdot(my, G3);
    e.a1c = e.dc - e.a2c - a3c; e.a1x = e.dx - e.a2x - a3x; e.a1y = e.dy - e.a2y - a3y;

    // conservative support: alpha >= 1/255 needs ecc <= E (see render.hip); ecc <= E <=> a_k >= m = (1 - E) / 3
    const float t = 255.0f * op;
    float E = -1.0f;
    if (t >= 1.0f)
    {
        const float L = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(t);
        if (GAMMA1) E = __builtin_amdgcn_sqrtf(L);
        else E = (g2 < 1e-6f) ? 10.0f : pow_nonneg(L, 1.0f / g2);
        E = fminf(E * 1.0005f + 0.002f, 10.01f);
    }
    const float m = (1.0f - E) * (1.0f / 3.0f);
    const float dmin = e.dc + fminf(0.0f, 7.0f * e.dx) + fminf(0.0f, 7.0f * e.dy);
    const float dmax = e.dc + fmaxf(0.0f, 7.0f * e.dx) + fmaxf(0.0f, 7.0f * e.dy);
    bool ov = E > 0.0f;
    if (dmin > 0.0f || dmax < 0.0f) // Den keeps its sign over the quadrant: a_k >= m  <=>  s (N_k - m Den) >= 0
    {
        const float s = dmin > 0.0f ? 1.0f : -1.0f;
        const float slack = 1e-3f * fmaxf(fabsf(dmin), fabsf(dmax));
        const float f1 = s * (e.a1c - m * e.dc) + fmaxf(0.0f, 7.0f * s * (e.a1x - m * e.dx)) + fmaxf(0.0f, 7.0f * s * (e.a1y - m * e.dy));
        const float f2 = s * (e.a2c - m * e.dc) + fmaxf(0.0f, 7.0f * s * (e.a2x - m * e.dx)) + fmaxf(0.0f, 7.0f * s * (e.a2y - m * e.dy));
        const float f3 = s * (a3c - m * e.dc) + fmaxf(0.0f, 7.0f * s * (a3x - m * e.dx)) + fmaxf(0.0f, 7.0f * s * (a3y - m * e.dy));
        ov = ov && f1 >= -slack && f2 >= -slack && f3 >= -slack;
    }
    e.overlap = ov;
    return e;
}

// LDS row layouts (floats).  Forward: 20 per entry; backward: 28 per entry, the 20 reduced sums alias [0..19], id at [27].
//   [0..3] a1x a1y a1c a2x   [4..7] a2y a2c dx dy   [8..11] dc d0 op r   [12..15] g b nx ny   [16] nz
//   backward only: [17] 1/n.n   [18..20] F1 = n x (v3 - v2) / n.n   [21..23] F2 = n x (v1 - v3) / n.n   [24..26] v1   [27] id
constexpr int CS3F = 20, CS3B = 28;

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256) render3d_fwd_kernel(RenderArgs a, float tan_fovx, float tan_fovy,
                                                            const uint2 *__restrict__ ranges,
                                                            const uint32_t *__restrict__ point_list,
                                                            const float4 *__restrict__ rec, float *__restrict__ final_T,
                                                            uint32_t *__restrict__ n_contrib, float *__restrict__ out_feature,
                                                            float *__restrict__ out_depth, float *__restrict__ out_normal,
                                                            float *__restrict__ contrib_sum, float *__restrict__ contrib_max)
{
    __shared__ __attribute__((aligned(16))) float cst_all[4][64 * CS3F];
    __shared__ float stage_all[RICH ? 4 : 1][8][64];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly;
    // pixToProj(v, S) = (2 v - S + 1) / S   (R3D auxiliary.h:40-43)
    const float sx = tan_fovx * 2.0f / (float)a.W, sy = tan_fovy * 2.0f / (float)a.H;
    const V3 ray0 = {tan_fovx * ((2.0f * (float)X0 - (float)a.W + 1.0f) / (float)a.W),
                     tan_fovy * ((2.0f * (float)Y0 - (float)a.H + 1.0f) / (float)a.H), 1.0f};
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    const float g2 = 2.0f * a.gamma;
    const float bg0 = a.background[0], bg1 = a.C > 1 ? a.background[1] : 0.0f, bg2 = a.C > 2 ? a.background[2] : 0.0f;
    float *cst = cst_all[wave];
    float(*stage)[64] = stage_all[RICH ? wave : 0];

    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, anx = 0.0f, any_ = 0.0f, anz = 0.0f, ad = 0.0f;
    bool done = !inside;
    uint32_t last = (uint32_t)len;

    int staged = 0;
    uint32_t staged_ids = 0;
    const int slot = RICH ? slot8_of_lane(lane) : 0;
    auto flush = [&]() { // contrib_sum / contrib_max, see render.hip
        float vs[8], vm[8];
#pragma unroll
        for (int i = 0; i < 8; i++)
        {
            const float x = (i < staged) ? stage[i][lane] : 0.0f;
            vs[i] = x;
            vm[i] = x;
        }
        const float rs = reduce8(vs, lane, OpAdd());
        const float rm = reduce8(vm, lane, OpMax());
        const uint32_t gid = (uint32_t)__shfl((int)staged_ids, slot);
        if ((lane & 7) == 0 && slot < staged)
        {
            unsafeAtomicAdd(contrib_sum + gid, rs);
            atomicMax((int *)contrib_max + gid, __float_as_int(rm));
        }
        staged = 0;
    };

    unsigned long long alive = __ballot(!done);
    for (int base = 0; base < len; base += 64)
    {
        if (alive == 0) break;
        const int k = base + lane;
        const bool valid = k < len;
        uint32_t id = 0;
        float4 r0 = make_float4(0, 0, 1, 0), r1 = make_float4(1, 0, 0, 1), r2 = make_float4(1, 0, 0, 1), r3 = make_float4(0, 0, 0, 0);
        if (valid)
        {
            id = point_list[range.x + k] & TS_ID_MASK; // id bits (the top four are a quadrant mask, ts2d_support.h)
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
        }
        const V3 v1 = {r0.x, r0.y, r0.z}, v2 = {r0.w, r1.x, r1.y}, v3 = {r1.z, r1.w, r2.x}, n = {r2.y, r2.z, r2.w};
        const Entry3 e = entry_setup3<GAMMA1>(v1, v2, v3, n, r3.x, g2, ray0, sx, sy);
        unsigned long long mask = __ballot(valid && e.overlap);
        if (mask == 0) continue;
        {
            float4 *q = (float4 *)(cst + lane * CS3F);
            q[0] = make_float4(e.a1x, e.a1y, e.a1c, e.a2x);
            q[1] = make_float4(e.a2y, e.a2c, e.dx, e.dy);
            q[2] = make_float4(e.dc, e.d0, r3.x, r3.y);
            q[3] = make_float4(r3.z, r3.w, n.x, n.y);
            cst[lane * CS3F + 16] = n.z;
        }
        while (mask)
        {
            const int jc = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float4 c0 = *(const float4 *)(cst + jc * CS3F), c1 = *(const float4 *)(cst + jc * CS3F + 4);
            const float4 c2 = *(const float4 *)(cst + jc * CS3F + 8);
            const float N1 = fmaf(c0.x, fx, fmaf(c0.y, fy, c0.z));
            const float N2 = fmaf(c0.w, fx, fmaf(c1.x, fy, c1.y));
            const float den = fmaf(c1.z, fx, fmaf(c1.w, fy, c2.x));
            const float inv = __builtin_amdgcn_rcpf(den);
            const float a1 = N1 * inv, a2 = N2 * inv, a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            bool hit = !done && fabsf(den) >= 1e-8f && ecc >= 0.0f && ecc <= 10.0f; // R3D forward.cu:241,256
            if (__ballot(hit) == 0) continue;
            const float pw = GAMMA1 ? ecc * ecc : pow_nonneg(ecc, g2);
            const float alpha = fminf(0.99f, c2.z * fast_exp(-0.5f * pw));
            hit = hit && alpha >= 1.0f / 255.0f; // R3D forward.cu:265
            if (__ballot(hit) == 0) continue;
            const float4 c3 = *(const float4 *)(cst + jc * CS3F + 12);
            const float al = hit ? alpha : 0.0f;
            const float contrib = al * T;
            ar = fmaf(c2.w, contrib, ar);
            ag = fmaf(c3.x, contrib, ag);
            ab = fmaf(c3.y, contrib, ab);
            if (RICH)
            {
                const float nz = cst[jc * CS3F + 16];
                anx = fmaf(c3.z, contrib, anx);
                any_ = fmaf(c3.w, contrib, any_);
                anz = fmaf(nz, contrib, anz);
                const float depth = hit ? c2.y * inv : 0.0f; // R3D forward.cu:244
                ad = fmaf(depth, contrib, ad);
                stage[staged][lane] = contrib;
                staged_ids = (lane == staged) ? bcast(id, jc) : staged_ids;
                if (++staged == 8) flush();
            }
            T *= (1.0f - al);
            if (hit && T <= 0.0001f)
            {
                done = true;
                last = (uint32_t)(base + jc + 1);
            }
            alive = __ballot(!done);
            if (alive == 0)
            {
                mask = 0;
                base = len;
            }
        }
    }
    if (RICH && staged > 0) flush();

    if (inside)
    {
        const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_feature[pix] = ar + T * bg0;
        if (a.C > 1) out_feature[HW + pix] = ag + T * bg1;
        if (a.C > 2) out_feature[2 * HW + pix] = ab + T * bg2;
        if (RICH)
        {
            out_depth[pix] = ad + T * (a.background_depth_dev ? *a.background_depth_dev : a.background_depth);
            out_normal[pix] = anx;
            out_normal[HW + pix] = any_;
            out_normal[2 * HW + pix] = anz;
        }
    }
}

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256) render3d_bwd_kernel(RenderArgs a, float tan_fovx, float tan_fovy,
                                                            const uint2 *__restrict__ ranges,
                                                            const uint32_t *__restrict__ point_list,
                                                            const float4 *__restrict__ rec, const float *__restrict__ final_T,
                                                            const uint32_t *__restrict__ n_contrib,
                                                            const float *__restrict__ dL_dout_feature,
                                                            const float *__restrict__ dL_dout_depth,
                                                            const float *__restrict__ dL_dout_normal, float *__restrict__ grad_rec)
{
    __shared__ __attribute__((aligned(16))) float cst_all[4][64 * CS3B];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly;
    const float sx = tan_fovx * 2.0f / (float)a.W, sy = tan_fovy * 2.0f / (float)a.H;
    const V3 ray0 = {tan_fovx * ((2.0f * (float)X0 - (float)a.W + 1.0f) / (float)a.W),
                     tan_fovy * ((2.0f * (float)Y0 - (float)a.H + 1.0f) / (float)a.H), 1.0f};
    const V3 ray = {tan_fovx * ((2.0f * (float)px - (float)a.W + 1.0f) / (float)a.W),
                    tan_fovy * ((2.0f * (float)py - (float)a.H + 1.0f) / (float)a.H), 1.0f}; // this pixel's p_ray
    const uint2 range = ranges[tile];
    const float g2 = 2.0f * a.gamma;
    const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
    float *cst = cst_all[wave];

    float T = inside ? final_T[pix] : 0.0f;
    const int last = inside ? (int)n_contrib[pix] : 0;
    // scalar back-to-front composite B, see render.hip
    float dpr = 0.0f, dpg = 0.0f, dpb = 0.0f, dnx = 0.0f, dny = 0.0f, dnz = 0.0f, dd = 0.0f, B = 0.0f;
    if (inside)
    {
        dpr = dL_dout_feature[pix];
        B = dpr * a.background[0];
        if (a.C > 1) { dpg = dL_dout_feature[HW + pix]; B = fmaf(dpg, a.background[1], B); }
        if (a.C > 2) { dpb = dL_dout_feature[2 * HW + pix]; B = fmaf(dpb, a.background[2], B); }
        if (RICH)
        {
            dnx = dL_dout_normal[pix]; dny = dL_dout_normal[HW + pix]; dnz = dL_dout_normal[2 * HW + pix];
            dd = dL_dout_depth[pix];
            B = fmaf(dd, a.background_depth_dev ? *a.background_depth_dev : a.background_depth, B);
        }
    }
    const int slot = slot_of_lane(lane), slot4 = slot4_of_lane(lane);
    const bool writer16 = (lane & 3) == 0, writer4 = (lane & 15) == 0;

    const int wlast = __builtin_amdgcn_readlane(__float_as_int(wave_max63_nonneg((float)last)), 63);
    const int maxlast = (int)__int_as_float(wlast);
    if (maxlast <= 0) return;

    for (int base = ((maxlast - 1) >> 6) << 6; base >= 0; base -= 64)
    {
        const int k = base + lane;
        const bool valid = k < maxlast;
        uint32_t id = 0;
        float4 r0 = make_float4(0, 0, 1, 0), r1 = make_float4(1, 0, 0, 1), r2 = make_float4(1, 0, 0, 1), r3 = make_float4(0, 0, 0, 0);
        if (valid)
        {
            id = point_list[range.x + k] & TS_ID_MASK; // id bits (the top four are a quadrant mask, ts2d_support.h)
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
        }
        const V3 ev1 = {r0.x, r0.y, r0.z}, ev2 = {r0.w, r1.x, r1.y}, ev3 = {r1.z, r1.w, r2.x}, en = {r2.y, r2.z, r2.w};
        // support from G >= 1/255 (opacity plays no part in the backward's skip test, R3D backward.cu:351)
        const Entry3 e = entry_setup3<GAMMA1>(ev1, ev2, ev3, en, 1.0f, g2, ray0, sx, sy);
        unsigned long long mask = __ballot(valid && e.overlap);
        if (mask == 0) continue;
        {
            float4 *q = (float4 *)(cst + lane * CS3B);
            q[0] = make_float4(e.a1x, e.a1y, e.a1c, e.a2x);
            q[1] = make_float4(e.a2y, e.a2c, e.dx, e.dy);
            q[2] = make_float4(e.dc, e.d0, r3.x, r3.y);
            q[3] = make_float4(r3.z, r3.w, en.x, en.y);
            const float inn_e = 1.0f / vdot(en, en);
            const V3 F1 = vscale(inn_e, vcross(en, vsub(ev3, ev2))), F2 = vscale(inn_e, vcross(en, vsub(ev1, ev3)));
            q[4] = make_float4(en.z, inn_e, F1.x, F1.y);
            q[5] = make_float4(F1.z, F2.x, F2.y, F2.z);
            q[6] = make_float4(ev1.x, ev1.y, ev1.z, __uint_as_float(id));
        }
        unsigned long long touched = 0;
        while (mask)
        {
            const int jc = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << jc);
            const float4 c0 = *(const float4 *)(cst + jc * CS3B), c1 = *(const float4 *)(cst + jc * CS3B + 4);
            const float4 c2 = *(const float4 *)(cst + jc * CS3B + 8);
            const float N1 = fmaf(c0.x, fx, fmaf(c0.y, fy, c0.z));
            const float N2 = fmaf(c0.w, fx, fmaf(c1.x, fy, c1.y));
            const float den = fmaf(c1.z, fx, fmaf(c1.w, fy, c2.x));
            const float inv = __builtin_amdgcn_rcpf(den);
            const float a1 = N1 * inv, a2 = N2 * inv, a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            bool hit = (base + jc < last) && fabsf(den) >= 1e-8f && ecc >= 0.0f && ecc <= 10.0f; // R3D backward.cu:322,326,340
            if (__ballot(hit) == 0) continue;
            const float pw = GAMMA1 ? ecc * ecc : pow_nonneg(ecc, g2);
            const float power = -0.5f * pw;
            const float op = c2.z;
            const float G = fast_exp(power);
            const float alpha = fminf(0.99f, op * G);
            hit = hit && G >= 1.0f / 255.0f; // sic: G, R3D backward.cu:351
            if (__ballot(hit) == 0) continue;

            const float4 c3 = *(const float4 *)(cst + jc * CS3B + 12), c4 = *(const float4 *)(cst + jc * CS3B + 16);
            const float4 c5 = *(const float4 *)(cst + jc * CS3B + 20);
            const float4 c6 = *(const float4 *)(cst + jc * CS3B + 24); // .w is the id, unused here
            const V3 n = {c3.z, c3.w, c4.x}, F1 = {c4.z, c4.w, c5.x}, F2 = {c5.y, c5.z, c5.w}, v1 = {c6.x, c6.y, c6.z};
            const float inn = c4.y, d0 = c2.y;

            const float al = hit ? alpha : 0.0f;
            const float oma = 1.0f - al;
            T = T * __builtin_amdgcn_rcpf(oma); // R3D backward.cu:354
            const float contrib = al * T;
            const float depth = hit ? d0 * inv : 0.0f;
            float X = fmaf(dpb, c3.y, fmaf(dpg, c3.x, dpr * c2.w)); // R3D backward.cu:368
            float dL_ddepth = 0.0f;
            if (RICH) // R3D backward.cu:373-382
            {
                X = fmaf(dnz, n.z, fmaf(dny, n.y, fmaf(dnx, n.x, X)));
                X = fmaf(dd, depth, X);
                dL_ddepth = dd * contrib;
            }
            const float dL_dcontrib = X - B;
            B = fmaf(al, X, oma * B);
            const float dL_dalpha = dL_dcontrib * T;
            const float decc_raw = dL_dalpha * alpha * g2 * power * __builtin_amdgcn_rcpf(ecc + 1e-8f); // R3D backward.cu:384-386
            const float z = (hit && op * G < 0.99f) ? -3.0f * decc_raw : 0.0f;
            const bool k1 = a1 <= a2 && a1 <= a3;        // R3D backward.cu:388-401
            const bool k2 = !k1 && a2 <= a1 && a2 <= a3;
            const bool k3 = !(k1 || k2);
            // dL/da = z e_k and a3 = 1 - a1 - a2:  sum_k dL/da_k da_k/dx = z (w1 da1/dx + w2 da2/dx)
            const float t1 = z * ((k1 ? 1.0f : 0.0f) - (k3 ? 1.0f : 0.0f)), t2 = z * ((k2 ? 1.0f : 0.0f) - (k3 ? 1.0f : 0.0f));
            const float zw1 = t1 * inn, zw2 = t2 * inn;
            // da1/ddepth = n . cross(v3 - v2, p_ray) / n.n = p_ray . F1, da2/ddepth = p_ray . F2   (:407,413,421)
            dL_ddepth += t1 * vdot(ray, F1) + t2 * vdot(ray, F2);
            const float dip = hit ? dL_ddepth * inv : 0.0f; // dL_ddepth * inv_p_ray_dot_n (:422-423)
            // q = hit point - v1 = -p_v1; p_v2 = (v2 - v1) - q, p_v3 = (v3 - v1) - q.  Every gradient term is linear in (1, q):
            //   dL/dv1 = sum zw2 n x p_v3 + dip n                      = Z2 n x e3 - n x Q2 + Dp n            (:410,425)
            //   dL/dv2 = -sum zw1 n x p_v3                             = -Z1 n x e3 + n x Q1                  (:404,426)
            //   dL/dv3 = sum zw1 n x p_v2 + zw2 p_v1 x n               = Z1 n x e2 - n x Q1 + n x Q2          (:405,412,427)
            //   dL/dn  = sum dn c + zw1 (p_v2 x p_v3 - 2 a1 n) + zw2 (p_v3 x p_v1 - 2 a2 n) + dip p_v1
            //          = Nn + (Z1 - S2) n + (e3 - e2) x Q1 + Q2 x e3 - Qd                                      (:376,406,414,423,428)
            // with e_k = v_k - v1, n = e2 x e3, Z = sum zw, Q = sum zw q, Dp = sum dip, Qd = sum dip q, S2 = sum 2 (zw1 a1 + zw2 a2).
            const V3 q = {fmaf(depth, ray.x, -v1.x), fmaf(depth, ray.y, -v1.y), depth - v1.z};
            const V3 qq = hit ? q : V3{0.0f, 0.0f, 0.0f};
            float v[16];
            v[0] = zw1; v[1] = zw2;
            v[2] = zw1 * qq.x; v[3] = zw1 * qq.y; v[4] = zw1 * qq.z;
            v[5] = zw2 * qq.x; v[6] = zw2 * qq.y; v[7] = zw2 * qq.z;
            v[8] = dip; v[9] = dip * qq.x; v[10] = dip * qq.y; v[11] = dip * qq.z;
            v[12] = 2.0f * (zw1 * a1 + zw2 * a2);
            v[13] = hit ? dL_dalpha * G : 0.0f;      // R3D backward.cu:451
            v[14] = dpr * contrib; v[15] = dpg * contrib; // R3D backward.cu:365
            const float r16 = reduce16(v, lane);
            const float r4 = reduce4(dpb * contrib, dnx * contrib, dny * contrib, dnz * contrib);
            if (writer16) cst[jc * CS3B + slot] = r16; // the entry's row is dead except for the id in slot 27
            if (writer4) cst[jc * CS3B + 16 + slot4] = r4;
            touched |= 1ull << jc;
        }
        if (touched == 0) continue;
        if ((touched >> lane) & 1) // the owning lane turns the 20 sums into the 16 values of the gradient record
        {
            const float4 *sq = (const float4 *)(cst + lane * CS3B);
            const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3], s4 = sq[4];
            const float Z1 = s0.x, Z2 = s0.y, Dp = s2.x, S2 = s3.x;
            const V3 Q1 = {s0.z, s0.w, s1.x}, Q2 = {s1.y, s1.z, s1.w}, Qd = {s2.y, s2.z, s2.w}, Nn = {s4.y, s4.z, s4.w};
            const V3 e2 = vsub(ev2, ev1), e3 = vsub(ev3, ev1);
            const V3 ne3 = vcross(en, e3), ne2 = vcross(en, e2), nQ1 = vcross(en, Q1), nQ2 = vcross(en, Q2);
            const V3 gv1 = {Z2 * ne3.x - nQ2.x + Dp * en.x, Z2 * ne3.y - nQ2.y + Dp * en.y, Z2 * ne3.z - nQ2.z + Dp * en.z};
            const V3 gv2 = {nQ1.x - Z1 * ne3.x, nQ1.y - Z1 * ne3.y, nQ1.z - Z1 * ne3.z};
            const V3 gv3 = {Z1 * ne2.x - nQ1.x + nQ2.x, Z1 * ne2.y - nQ1.y + nQ2.y, Z1 * ne2.z - nQ1.z + nQ2.z};
            const V3 eQ1 = vcross(vsub(e3, e2), Q1), Qe3 = vcross(Q2, e3);
            const float zs = Z1 - S2;
            const V3 gn = {Nn.x + zs * en.x + eQ1.x + Qe3.x - Qd.x, Nn.y + zs * en.y + eQ1.y + Qe3.y - Qd.y,
                           Nn.z + zs * en.z + eQ1.z + Qe3.z - Qd.z};
            float4 *gq = (float4 *)(cst + lane * CS3B); // grad-record order: gv1 gv2 gv3 gn dL/dopacity dL/drgb
            gq[0] = make_float4(gv1.x, gv1.y, gv1.z, gv2.x);
            gq[1] = make_float4(gv2.y, gv2.z, gv3.x, gv3.y);
            gq[2] = make_float4(gv3.z, gn.x, gn.y, gn.z);
            gq[3] = make_float4(s3.y /* dL/dopacity */, s3.z, s3.w, s4.x /* dL/drgb */);
        }
        {
            const int sub = lane >> 4, col = lane & 15;
#pragma unroll 1
            for (int e0 = 0; e0 < 64; e0 += 4)
            {
                if (((touched >> e0) & 0xFull) == 0) continue;
                const int ee = e0 + sub;
                if ((touched >> ee) & 1)
                {
                    const uint32_t eid = __float_as_uint(cst[ee * CS3B + 27]);
                    unsafeAtomicAdd(grad_rec + TS_GRAD_FLOATS * (size_t)eid + col, cst[ee * CS3B + col]);
                }
            }
        }
    }
}
} // namespace

#define TS_DISPATCH3(KERNEL, ...)                                                                                     \
    do                                                                                                                \
    {                                                                                                                 \
        const bool g1 = (a.gamma == 1.0f);                                                                            \
        if (a.rich_info && g1) hipLaunchKernelGGL((KERNEL<true, true>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (a.rich_info) hipLaunchKernelGGL((KERNEL<true, false>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (g1) hipLaunchKernelGGL((KERNEL<false, true>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, dim3(256), 0, s, __VA_ARGS__);                          \
    } while (0)

void ts_launch_render3d_fwd(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                            const BinningStateView &b, const ImageStateView &im, float *out_feature, float *out_depth,
                            float *out_normal, float *contrib_sum, float *contrib_max, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y ? ts_tile_units(a.grid_x, a.grid_y) : 0));
    if (grid.x == 0) return;
    TS_DISPATCH3(render3d_fwd_kernel, a, tan_fovx, tan_fovy, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature,
                 out_depth, out_normal, contrib_sum, contrib_max);
}

void ts_launch_render3d_bwd(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                            const BinningStateView &b, const ImageStateView &im, const float *dL_dout_feature,
                            const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y ? ts_tile_units(a.grid_x, a.grid_y) : 0));
    if (grid.x == 0) return;
    TS_DISPATCH3(render3d_bwd_kernel, a, tan_fovx, tan_fovy, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib,
                 dL_dout_feature, dL_dout_depth, dL_dout_normal, grad_rec);
}
