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
//   * the backward's 16 values are already the 64-byte gradient record (dL/dv1_view, dv2_view, dv3_view, dnormal_view,
//     dopacity, drgb), so the batch epilogue only flushes.
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

// LDS row layouts (floats).  Forward: 20 per entry; backward: 28 per entry, sums alias [0..15], id at [27].
//   [0..3] a1x a1y a1c a2x   [4..7] a2y a2c dx dy   [8..11] dc d0 op r   [12..15] g b nx ny   [16] nz
//   backward only: [17..19] v1   [20..22] v2   [23..25] v3   [27] id
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

    const int tile = tile_of_block(blockIdx.x, a.grid_x * a.grid_y);
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
            id = point_list[range.x + k];
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
            out_depth[pix] = ad + T * a.background_depth;
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

    const int tile = tile_of_block(blockIdx.x, a.grid_x * a.grid_y);
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
            B = fmaf(dd, a.background_depth, B);
        }
    }
    const int slot = slot_of_lane(lane);
    const bool writer16 = (lane & 3) == 0;

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
            id = point_list[range.x + k];
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
            q[4] = make_float4(en.z, ev1.x, ev1.y, ev1.z);
            q[5] = make_float4(ev2.x, ev2.y, ev2.z, ev3.x);
            q[6] = make_float4(ev3.y, ev3.z, 0.0f, __uint_as_float(id));
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
            const float2 c6 = *(const float2 *)(cst + jc * CS3B + 24);
            const V3 n = {c3.z, c3.w, c4.x}, v1 = {c4.y, c4.z, c4.w}, v2 = {c5.x, c5.y, c5.z}, v3 = {c5.w, c6.x, c6.y};
            const float d0 = c2.y;

            const float al = hit ? alpha : 0.0f;
            const float oma = 1.0f - al;
            T = T * __builtin_amdgcn_rcpf(oma); // R3D backward.cu:354
            const float contrib = al * T;
            const float depth = d0 * inv;
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
            const float w1 = (k1 ? 1.0f : 0.0f) - (k3 ? 1.0f : 0.0f), w2 = (k2 ? 1.0f : 0.0f) - (k3 ? 1.0f : 0.0f);
            const float inn = __builtin_amdgcn_rcpf(vdot(n, n));
            const float zw1 = z * w1 * inn, zw2 = z * w2 * inn;
            const V3 p = vscale(depth, ray);
            const V3 p1 = vsub(v1, p), p2 = vsub(v2, p), p3 = vsub(v3, p);
            // da1/ddepth = n . cross(v3 - v2, p_ray) / n.n ; da2/ddepth = n . cross(v1 - v3, p_ray) / n.n  (:407,413)
            const float da1_dd = vdot(n, vcross(vsub(v3, v2), ray)), da2_dd = vdot(n, vcross(vsub(v1, v3), ray));
            dL_ddepth += zw1 * da1_dd + zw2 * da2_dd; // R3D backward.cu:421
            const V3 c_n3 = vcross(n, p3), c_n2 = vcross(n, p2), c_1n = vcross(p1, n), c_23 = vcross(p2, p3), c_31 = vcross(p3, p1);
            const float dip = hit ? dL_ddepth * inv : 0.0f; // dL_ddepth * inv_p_ray_dot_n (:422-423)
            float v[16];
            // dL/dv1_view = dL_da.y da2_dv1 + dL_da.z da3_dv1 + dL_ddepth ddepth_dv1, da2_dv1 = cross(n, p_v3)/n.n (:410,425)
            v[0] = fmaf(zw2, c_n3.x, dip * n.x); v[1] = fmaf(zw2, c_n3.y, dip * n.y); v[2] = fmaf(zw2, c_n3.z, dip * n.z);
            // dL/dv2_view: da1_dv2 = cross(p_v3, n)/n.n = -cross(n, p_v3)/n.n (:404,426)
            v[3] = -zw1 * c_n3.x; v[4] = -zw1 * c_n3.y; v[5] = -zw1 * c_n3.z;
            // dL/dv3_view: da1_dv3 = cross(n, p_v2)/n.n, da2_dv3 = cross(p_v1, n)/n.n (:405,412,427)
            v[6] = fmaf(zw1, c_n2.x, zw2 * c_1n.x); v[7] = fmaf(zw1, c_n2.y, zw2 * c_1n.y); v[8] = fmaf(zw1, c_n2.z, zw2 * c_1n.z);
            // dL/dnormal_view (:376,406,414,423,428): da_k/dn = (cross(..) - 2 a_k n)/n.n, ddepth/dn = (v1 - depth p_ray)/(p_ray.n)
            const float s2 = 2.0f * (zw1 * a1 + zw2 * a2);
            v[9] = fmaf(dnx, contrib, fmaf(zw1, c_23.x, fmaf(zw2, c_31.x, fmaf(dip, p1.x, -s2 * n.x))));
            v[10] = fmaf(dny, contrib, fmaf(zw1, c_23.y, fmaf(zw2, c_31.y, fmaf(dip, p1.y, -s2 * n.y))));
            v[11] = fmaf(dnz, contrib, fmaf(zw1, c_23.z, fmaf(zw2, c_31.z, fmaf(dip, p1.z, -s2 * n.z))));
            v[12] = hit ? dL_dalpha * G : 0.0f; // R3D backward.cu:451
            v[13] = dpr * contrib; v[14] = dpg * contrib; v[15] = dpb * contrib; // R3D backward.cu:365
            if (!RICH) { /* dnx.. and dd are zero: same expressions */ }
            const float r16 = reduce16(v, lane);
            if (writer16) cst[jc * CS3B + slot] = r16; // the entry's row is dead except for the id in slot 27
            touched |= 1ull << jc;
        }
        if (touched == 0) continue;
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
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    TS_DISPATCH3(render3d_fwd_kernel, a, tan_fovx, tan_fovy, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature,
                 out_depth, out_normal, contrib_sum, contrib_max);
}

void ts_launch_render3d_bwd(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                            const BinningStateView &b, const ImageStateView &im, const float *dL_dout_feature,
                            const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    TS_DISPATCH3(render3d_bwd_kernel, a, tan_fovx, tan_fovy, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib,
                 dL_dout_feature, dL_dout_depth, dL_dout_normal, grad_rec);
}
