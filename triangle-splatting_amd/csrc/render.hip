// render.hip -- per-pixel alpha blend (forward) and its back-to-front replay (backward).
//
// Behaviour follows FORWARD::renderCUDA (R2D/src/forward.cu:198-355) and BACKWARD::renderCUDA
// (R2D/src/backward.cu:265-493); Appendix B of SURVEY.md lists the quirks that are kept (integer pixel
// centres, n_contrib counts examined entries, stop AFTER the triangle that drives T <= 1e-4, dL_dopacity not
// gated by the 0.99 clamp, arg-min tie order a1, a2, a3, division by ecc + 1e-8).
//
// Structure is CDNA4-first and differs from the reference's (one 16x16 thread block per tile, 256-entry
// shared-memory batches, two __syncthreads per batch, per-(pixel,triangle) global atomics):
//
//   * one wave64 per 8x8 pixel quadrant; the four quadrant waves of a tile form one 256-thread workgroup so
//     their record gathers share the CU's L1, but they never synchronise (no LDS, no barrier) and each
//     terminates as soon as its own 64 pixels are saturated;
//   * a batch = 64 list entries, one per lane: each lane gathers its entry's 64-byte record with four dwordx4
//     loads, does the per-(entry,quadrant) setup once (edge functions as affine forms of the in-quadrant pixel
//     offset, conservative support box) and the wave ballots the entries whose support box meets the quadrant;
//   * the wave then walks the set bits of that ballot (s_ff1 / s_flbit); per-entry constants reach all lanes
//     through v_readlane (SGPR broadcast), and the per-pixel test is 4 FMAs + min3 + 2 compares;
//   * backward: the 16 per-triangle gradient terms of the 64 pixels are reduced inside the wave by a
//     transpose-reduce network (v_permlane32_swap / v_permlane16_swap / DPP, 35 VALU ops instead of 16 x 6 for
//     independent butterflies) that leaves each of the 16 sums in a distinct lane quad, so one 16-lane
//     global_atomic_add_f32 on the triangle's 64-byte gradient record replaces the reference's 16 x 64
//     atomics per (tile quadrant, triangle).
//
// Skipping entries by the support box cannot change results: an entry is only skipped for a quadrant when no
// pixel of the quadrant can pass the reference's own tests (0 <= ecc <= 10 and alpha >= 1/255), and
// n_contrib / termination are tracked by list position exactly as the reference counts them.
#include "ts2d_common.h"

namespace
{
__device__ __forceinline__ float bcast(float v, int j)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
constexpr int DPP_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROR8 = 0x128;       // row_ror:8  (lane ^ 8 inside a row of 16)
constexpr int DPP_HALF_MIRROR = 0x141; // lane -> 7 - lane inside each group of 8
constexpr int DPP_MIRROR = 0x140;     // lane -> 15 - lane inside a row of 16
constexpr int DPP_BCAST15 = 0x142;    // lane 15 of row r -> all lanes of row r+1
constexpr int DPP_BCAST31 = 0x143;    // lane 31 -> all lanes of rows 2,3

// Full 64-lane reductions; result valid in lane 63 (read back with bcast(v, 63)).
__device__ __forceinline__ float wave_sum63(float v)
{
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    v += dpp<DPP_HALF_MIRROR>(v);
    v += dpp<DPP_MIRROR>(v);
    v += dpp<DPP_BCAST15, 0xA>(v);
    v += dpp<DPP_BCAST31, 0xC>(v);
    return v;
}
__device__ __forceinline__ float wave_max63_nonneg(float v) // inputs >= 0 (masked-off rows contribute 0)
{
    v = fmaxf(v, dpp<DPP_XOR1>(v));
    v = fmaxf(v, dpp<DPP_XOR2>(v));
    v = fmaxf(v, dpp<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp<DPP_MIRROR>(v));
    v = fmaxf(v, dpp<DPP_BCAST15, 0xA>(v));
    v = fmaxf(v, dpp<DPP_BCAST31, 0xC>(v));
    return v;
}

__device__ __forceinline__ void swap32(float &a, float &b) // a <- [a.lo | b.lo], b <- [a.hi | b.hi]
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float &a, float &b) // odd rows of a <-> even rows of b
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// Transpose-reduce: 16 values per lane x 64 lanes -> each lane returns the complete 64-lane reduction of ONE of
// the 16 values; the four lanes of a quad hold the same value and the 16 quads hold the 16 different values.
// Which value a lane ends up with is discovered once per wave by reducing indicator inputs (slot_of_lane()).
struct OpAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };

template <typename Op>
__device__ __forceinline__ float reduce16(float (&v)[16], int lane, Op op)
{
#pragma unroll
    for (int i = 0; i < 8; i++) // 64 -> 32 lanes per value, two values per register
    {
        swap32(v[2 * i], v[2 * i + 1]);
        v[i] = op(v[2 * i], v[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) // 32 -> 16 lanes per value, one value per row
    {
        swap16(v[2 * i], v[2 * i + 1]);
        v[i] = op(v[2 * i], v[2 * i + 1]);
    }
    const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 2; i++) // 16 -> 8 lanes per value
    {
        const float own = b3 ? v[2 * i + 1] : v[2 * i];
        const float oth = b3 ? v[2 * i] : v[2 * i + 1];
        v[i] = op(own, dpp<DPP_ROR8>(oth));
    }
    {
        const float own = b2 ? v[1] : v[0]; // 8 -> 4 lanes per value
        const float oth = b2 ? v[0] : v[1];
        v[0] = op(own, dpp<DPP_HALF_MIRROR>(oth));
    }
    float r = v[0];
    r = op(r, dpp<DPP_XOR1>(r));
    r = op(r, dpp<DPP_XOR2>(r));
    return r;
}
__device__ __forceinline__ float reduce16(float (&v)[16], int lane) { return reduce16(v, lane, OpAdd()); }

__device__ __forceinline__ int slot_of_lane(int lane)
{
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = (lane == 0) ? (float)i : 0.0f;
    return (int)reduce16(v, lane);
}

// x^y for x >= 0, y >= 0 via v_log_f32 / v_exp_f32 (x = 0 -> 0, y = 0 -> 1 like powf).
__device__ __forceinline__ float pow_nonneg(float x, float y)
{
    const float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
    return y == 0.0f ? 1.0f : r;
}
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// blockIdx -> tile so that each XCD (block b runs on XCD b % 8) owns one contiguous band of row-major tiles:
// neighbouring tiles gather mostly the same triangle records, which then stay in that XCD's 4 MiB L2.
__device__ __forceinline__ int tile_of_block(int b, int ntiles)
{
    const int q = ntiles >> 3, r = ntiles & 7, x = b & 7, i = b >> 3;
    return x * q + min(x, r) + i;
}

// Per-entry setup shared by forward and backward.  All values live in the lane that owns the entry.
struct EntrySetup
{
    float A1, B1, C1, A2, B2, C2; // a1(q) = A1*qx + B1*qy + C1 (q = pixel offset inside the quadrant), same for a2
    bool overlap;                 // support box meets the quadrant
};

template <bool GAMMA1>
__device__ __forceinline__ EntrySetup entry_setup(float v1x, float v1y, float v2x, float v2y, float v3x, float v3y,
                                                  float op, float g2, float OX, float OY, float &inv_area,
                                                  float &u1x, float &u1y, float &u2x, float &u2y, float &u3x, float &u3y)
{
    EntrySetup s;
    const float area2 = (v2x - v1x) * (v3y - v1y) - (v2y - v1y) * (v3x - v1x); // the value the reference stores, forward.cu:137
    inv_area = 1.0f / area2;
    u1x = v1x - OX; u1y = v1y - OY; u2x = v2x - OX; u2y = v2y - OY; u3x = v3x - OX; u3y = v3y - OY;
    s.C1 = (u2x * u3y - u2y * u3x) * inv_area;
    s.A1 = (v2y - v3y) * inv_area;
    s.B1 = (v3x - v2x) * inv_area;
    s.C2 = (u3x * u1y - u3y * u1x) * inv_area;
    s.A2 = (v3y - v1y) * inv_area;
    s.B2 = (v1x - v3x) * inv_area;
    // Conservative support: alpha >= 1/255 needs ecc^(2 gamma) <= 2 ln(255 op), and ecc <= E is the triangle
    // scaled by E about its centroid.
    const float t = 255.0f * op;
    float E = -1.0f;
    if (t >= 1.0f)
    {
        const float L = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(t);
        if (GAMMA1) E = __builtin_amdgcn_sqrtf(L);
        else E = (g2 < 1e-6f) ? 10.0f : pow_nonneg(L, 1.0f / g2);
        E = fminf(E * 1.0005f + 0.002f, 10.01f);
    }
    const float cx = (u1x + u2x + u3x) * (1.0f / 3.0f), cy = (u1y + u2y + u3y) * (1.0f / 3.0f);
    const float e1x = E * (u1x - cx), e2x = E * (u2x - cx), e3x = E * (u3x - cx);
    const float e1y = E * (u1y - cy), e2y = E * (u2y - cy), e3y = E * (u3y - cy);
    const float pad = 0.05f;
    const float bminx = cx + fminf(fminf(e1x, e2x), e3x) - pad, bmaxx = cx + fmaxf(fmaxf(e1x, e2x), e3x) + pad;
    const float bminy = cy + fminf(fminf(e1y, e2y), e3y) - pad, bmaxy = cy + fmaxf(fmaxf(e1y, e2y), e3y) + pad;
    // Separating-axis test of the E-scaled triangle against the quadrant's 8x8 sample box: box axes (the bbox
    // above) plus the three edge normals.  ecc <= E  <=>  min_i a_i >= (1 - E) / 3, and each a_i is affine in q,
    // so its maximum over the box is C_i + max(0, 7 A_i) + max(0, 7 B_i).
    const float m = (1.0f - E) * (1.0f / 3.0f);
    const float A3 = -s.A1 - s.A2, B3 = -s.B1 - s.B2, C3 = 1.0f - s.C1 - s.C2;
    const float max1 = s.C1 + fmaxf(0.0f, 7.0f * s.A1) + fmaxf(0.0f, 7.0f * s.B1);
    const float max2 = s.C2 + fmaxf(0.0f, 7.0f * s.A2) + fmaxf(0.0f, 7.0f * s.B2);
    const float max3 = C3 + fmaxf(0.0f, 7.0f * A3) + fmaxf(0.0f, 7.0f * B3);
    s.overlap = (E > 0.0f) && bminx <= 7.0f && bmaxx >= 0.0f && bminy <= 7.0f && bmaxy >= 0.0f && max1 >= m &&
                max2 >= m && max3 >= m;
    return s;
}

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256) render_fwd_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                          const uint32_t *__restrict__ point_list,
                                                          const float4 *__restrict__ rec, float *__restrict__ final_T,
                                                          uint32_t *__restrict__ n_contrib, float *__restrict__ out_feature,
                                                          float *__restrict__ out_depth, float *__restrict__ out_normal,
                                                          float *__restrict__ contrib_sum, float *__restrict__ contrib_max)
{
    const int tile = tile_of_block(blockIdx.x, a.grid_x * a.grid_y);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly, OX = (float)X0, OY = (float)Y0;
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    const float g2 = 2.0f * a.gamma;
    const float bg0 = a.background[0], bg1 = a.C > 1 ? a.background[1] : 0.0f, bg2 = a.C > 2 ? a.background[2] : 0.0f;

    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, anx = 0.0f, any_ = 0.0f, anz = 0.0f, ad = 0.0f;
    bool done = !inside;
    uint32_t last = (uint32_t)len; // a pixel that never saturates examines the whole list (forward.cu:296-297)

    // contrib_sum / contrib_max (forward.cu:323-324): the reference issues two global atomics per (pixel, triangle).
    // Here each contributing entry parks its 64 per-pixel contributions in a wave-private LDS slot; every 16
    // entries the 16 x 64 block is reduced by two transpose-reduce passes (sum, max) and leaves as ONE 16-lane
    // atomic add + ONE 16-lane atomic max.  (LDS float atomics are not an option: ds_add_f32 measures ~190
    // cycles per wave instruction on gfx950, see profiles/r01_lds_atomic_microbench.txt.)
    __shared__ float stage[4][16][64];
    int staged = 0;                 // entries parked so far (wave-uniform)
    uint32_t staged_ids = 0;        // lane k holds the triangle id of parked entry k
    const int slot = RICH ? slot_of_lane(lane) : 0;
    auto flush = [&]() {
        float vs[16], vm[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const float x = (i < staged) ? stage[wave][i][lane] : 0.0f;
            vs[i] = x;
            vm[i] = x;
        }
        const float rs = reduce16(vs, lane, OpAdd());
        const float rm = reduce16(vm, lane, OpMax());
        const uint32_t gid = (uint32_t)__shfl((int)staged_ids, slot);
        if ((lane & 3) == 0 && slot < staged)
        {
            unsafeAtomicAdd(contrib_sum + gid, rs);
            atomicMax((int *)contrib_max + gid, __float_as_int(rm)); // rm > 0: int order == float order
        }
        staged = 0;
    };

    for (int base = 0; base < len; base += 64)
    {
        if (__ballot(!done) == 0) break;
        const int k = base + lane;
        const bool valid = k < len;
        uint32_t id = 0;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (valid)
        {
            id = point_list[range.x + k];
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            if (RICH) r3 = rp[3];
        }
        float inv_area, u1x, u1y, u2x, u2y, u3x, u3y;
        const EntrySetup s = entry_setup<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY, inv_area, u1x, u1y,
                                                 u2x, u2y, u3x, u3y);
        unsigned long long mask = __ballot(valid && s.overlap);
        while (mask)
        {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float a1 = fmaf(bcast(s.A1, j), fx, fmaf(bcast(s.B1, j), fy, bcast(s.C1, j)));
            const float a2 = fmaf(bcast(s.A2, j), fx, fmaf(bcast(s.B2, j), fy, bcast(s.C2, j)));
            const float a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            bool hit = !done && ecc >= 0.0f && ecc <= 10.0f; // forward.cu:307
            if (__ballot(hit) == 0) continue;
            const float pw = GAMMA1 ? ecc * ecc : pow_nonneg(ecc, g2);
            const float alpha = fminf(0.99f, bcast(r1.z, j) * fast_exp(-0.5f * pw)); // forward.cu:311-312
            hit = hit && alpha >= 1.0f / 255.0f;                                       // forward.cu:313
            if (__ballot(hit) == 0) continue;
            // Branch-free blend: lanes that do not hit run with alpha = 0, which leaves every accumulator and T
            // bit-unchanged (x + c*0 == x, T*1 == T).
            const float al = hit ? alpha : 0.0f;
            const float contrib = al * T;
            ar = fmaf(bcast(r1.w, j), contrib, ar);
            ag = fmaf(bcast(r2.x, j), contrib, ag);
            ab = fmaf(bcast(r2.y, j), contrib, ab);
            if (RICH)
            {
                anx = fmaf(bcast(r2.z, j), contrib, anx);
                any_ = fmaf(bcast(r2.w, j), contrib, any_);
                anz = fmaf(bcast(r3.x, j), contrib, anz);
                const float d = bcast(r3.y, j) * a1 + bcast(r3.z, j) * a2 + bcast(r3.w, j) * a3; // forward.cu:328
                ad = fmaf(d, contrib, ad);
                stage[wave][staged][lane] = contrib;
                staged_ids = (lane == staged) ? bcast(id, j) : staged_ids;
                if (++staged == 16) flush();
            }
            T *= (1.0f - al);
            if (hit && T <= 0.0001f) // forward.cu:333
            {
                done = true;
                last = (uint32_t)(base + j + 1);
            }
            if (__ballot(!done) == 0)
            {
                mask = 0;
                base = len; // leave both loops
            }
        }
    }
    if (RICH && staged > 0) flush();

    if (inside)
    {
        const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_feature[pix] = ar + T * bg0; // forward.cu:345
        if (a.C > 1) out_feature[HW + pix] = ag + T * bg1;
        if (a.C > 2) out_feature[2 * HW + pix] = ab + T * bg2;
        if (RICH)
        {
            out_depth[pix] = ad + T * a.background_depth; // forward.cu:349
            out_normal[pix] = anx;
            out_normal[HW + pix] = any_;
            out_normal[2 * HW + pix] = anz;
        }
    }
}

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256) render_bwd_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                          const uint32_t *__restrict__ point_list,
                                                          const float4 *__restrict__ rec, const float *__restrict__ final_T,
                                                          const uint32_t *__restrict__ n_contrib,
                                                          const float *__restrict__ dL_dout_feature,
                                                          const float *__restrict__ dL_dout_depth,
                                                          const float *__restrict__ dL_dout_normal, float *__restrict__ grad_rec)
{
    const int tile = tile_of_block(blockIdx.x, a.grid_x * a.grid_y);
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = lane & 7, ly = lane >> 3;
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly, OX = (float)X0, OY = (float)Y0;
    const uint2 range = ranges[tile];
    const float g2 = 2.0f * a.gamma;
    const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;

    float T = inside ? final_T[pix] : 0.0f;                  // backward.cu:318
    const int last = inside ? (int)n_contrib[pix] : 0;       // backward.cu:320
    // The reference keeps seven back-to-front composites per pixel (accum_feature[3], accum_normal, accum_depth,
    // backward.cu:323-325) but only ever uses them through dL_dcontrib = sum_c dL_dpix_c * (value_c - accum_c)
    // (:415,425,435).  With X = sum_c dL_dpix_c * value_c and B = sum_c dL_dpix_c * accum_c this is X - B, and the
    // per-channel update accum_c <- alpha*value_c + (1-alpha)*accum_c collapses to B <- alpha*X + (1-alpha)*B:
    // one scalar of sequential state instead of seven (same mathematics, different rounding order).
    float dpr = 0.0f, dpg = 0.0f, dpb = 0.0f, dnx = 0.0f, dny = 0.0f, dnz = 0.0f, dd = 0.0f, B = 0.0f;
    if (inside) // backward.cu:331-343
    {
        dpr = dL_dout_feature[pix];
        B = dpr * a.background[0];
        if (a.C > 1) { dpg = dL_dout_feature[HW + pix]; B = fmaf(dpg, a.background[1], B); }
        if (a.C > 2) { dpb = dL_dout_feature[2 * HW + pix]; B = fmaf(dpb, a.background[2], B); }
        if (RICH)
        {
            dnx = dL_dout_normal[pix]; dny = dL_dout_normal[HW + pix]; dnz = dL_dout_normal[2 * HW + pix];
            dd = dL_dout_depth[pix];
            B = fmaf(dd, a.background_depth, B); // accum_normal starts at 0, accum_depth at background_depth
        }
    }
    const int slot = slot_of_lane(lane);
    const bool writer = ((lane & 3) == 0) && (RICH || slot < 10);

    // entries at list positions >= max(last) are skipped by every pixel of the quadrant (backward.cu:377-379)
    const int wlast = __builtin_amdgcn_readlane(__float_as_int(wave_max63_nonneg((float)last)), 63);
    const int maxlast = (int)__int_as_float(wlast);
    if (maxlast <= 0) return;

    for (int base = ((maxlast - 1) >> 6) << 6; base >= 0; base -= 64)
    {
        const int k = base + lane;
        const bool valid = k < maxlast;
        uint32_t id = 0;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (valid)
        {
            id = point_list[range.x + k];
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            if (RICH) r3 = rp[3];
        }
        float inv_area, u1x, u1y, u2x, u2y, u3x, u3y;
        const EntrySetup s = entry_setup<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY, inv_area, u1x, u1y,
                                                 u2x, u2y, u3x, u3y);
        unsigned long long mask = __ballot(valid && s.overlap);
        while (mask)
        {
            const int j = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << j);
            const float sA1 = bcast(s.A1, j), sB1 = bcast(s.B1, j), sA2 = bcast(s.A2, j), sB2 = bcast(s.B2, j);
            const float a1 = fmaf(sA1, fx, fmaf(sB1, fy, bcast(s.C1, j)));
            const float a2 = fmaf(sA2, fx, fmaf(sB2, fy, bcast(s.C2, j)));
            const float a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            bool hit = (base + j < last) && ecc >= 0.0f && ecc <= 10.0f; // backward.cu:378,393
            if (__ballot(hit) == 0) continue;
            const float pw = GAMMA1 ? ecc * ecc : pow_nonneg(ecc, g2);
            const float power = -0.5f * pw;
            const float op = bcast(r1.z, j);
            const float G = fast_exp(power);
            const float alpha = fminf(0.99f, op * G);
            hit = hit && alpha >= 1.0f / 255.0f; // backward.cu:400
            if (__ballot(hit) == 0) continue;

            // Branch-free from here on: lanes that do not hit run with alpha = 0 so that T, B stay bit-unchanged
            // and every gradient term they produce is an exact 0 (all terms carry a factor alpha or contrib).
            float v[16];
            const float al = hit ? alpha : 0.0f;
            const float oma = 1.0f - al;
            T = T * __builtin_amdgcn_rcpf(oma); // backward.cu:403
            const float contrib = al * T;
            const float fr = bcast(r1.w, j), fg = bcast(r2.x, j), fb = bcast(r2.y, j);
            v[7] = dpr * contrib; v[8] = dpg * contrib; v[9] = dpb * contrib; // backward.cu:412
            float X = fmaf(dpb, fb, fmaf(dpg, fg, dpr * fr));
            float da1 = 0.0f, da2 = 0.0f, da3 = 0.0f;
            if (RICH) // backward.cu:419-437
            {
                const float nx = bcast(r2.z, j), ny = bcast(r2.w, j), nz = bcast(r3.x, j);
                v[10] = dnx * contrib; v[11] = dny * contrib; v[12] = dnz * contrib;
                X = fmaf(dnz, nz, fmaf(dny, ny, fmaf(dnx, nx, X)));
                const float dL_ddepth = dd * contrib;
                v[13] = dL_ddepth * a1; v[14] = dL_ddepth * a2; v[15] = dL_ddepth * a3;
                const float vd1 = bcast(r3.y, j), vd2 = bcast(r3.z, j), vd3 = bcast(r3.w, j);
                da1 = dL_ddepth * vd1; da2 = dL_ddepth * vd2; da3 = dL_ddepth * vd3;
                const float depth = fmaf(vd3, a3, fmaf(vd2, a2, vd1 * a1));
                X = fmaf(dd, depth, X);
            }
            else
            {
                v[10] = v[11] = v[12] = v[13] = v[14] = v[15] = 0.0f;
            }
            const float dL_dcontrib = X - B;
            B = fmaf(al, X, oma * B);
            const float dL_dalpha = dL_dcontrib * T;
            // backward.cu:443-447: dL_decc = dL_dpower * 2 gamma * power / (ecc + 1e-8), dL_dpower = dL_dalpha * alpha
            // unless the 0.99 clamp was active.  The select sits last so that a non-hit lane never multiplies 0 * inf.
            const float decc_raw = dL_dalpha * alpha * g2 * power * __builtin_amdgcn_rcpf(ecc + 1e-8f);
            const float z = (hit && op * G < 0.99f) ? -3.0f * decc_raw : 0.0f;
            if (a1 <= a2 && a1 <= a3) da1 += z; // backward.cu:449-461 (ties: a1, then a2)
            else if (a2 <= a1 && a2 <= a3) da2 += z;
            else da3 += z;
            // backward.cu:464-479 regrouped: with E_k = perp(opposite edge of vertex k) / area2 = -(A_k, B_k) and
            // S = sum_i dL/da_i * a_i:  dL/dv1 = S*E1 + perp(da3*p_v2 - da2*p_v3)/area2, cyclically for v2, v3.
            const float S = da1 * a1 + da2 * a2 + da3 * a3;
            const float ia = bcast(inv_area, j);
            const float p1x = bcast(u1x, j) - fx, p1y = bcast(u1y, j) - fy;
            const float p2x = bcast(u2x, j) - fx, p2y = bcast(u2y, j) - fy;
            const float p3x = bcast(u3x, j) - fx, p3y = bcast(u3y, j) - fy;
            const float sA3 = -sA1 - sA2, sB3 = -sB1 - sB2;
            const float t1x = da3 * p2x - da2 * p3x, t1y = da3 * p2y - da2 * p3y;
            const float t2x = da1 * p3x - da3 * p1x, t2y = da1 * p3y - da3 * p1y;
            const float t3x = da2 * p1x - da1 * p2x, t3y = da2 * p1y - da1 * p2y;
            v[0] = ia * t1y - S * sA1; v[1] = -ia * t1x - S * sB1;
            v[2] = ia * t2y - S * sA2; v[3] = -ia * t2x - S * sB2;
            v[4] = ia * t3y - S * sA3; v[5] = -ia * t3x - S * sB3;
            v[6] = hit ? dL_dalpha * G : 0.0f; // backward.cu:490 (not gated by the clamp)
            const float r = reduce16(v, lane);
            if (writer) unsafeAtomicAdd(grad_rec + TS_GRAD_FLOATS * (size_t)bcast(id, j) + slot, r);
        }
    }
}
} // namespace

#define TS_DISPATCH(KERNEL, ...)                                                                                      \
    do                                                                                                                \
    {                                                                                                                 \
        const bool g1 = (a.gamma == 1.0f);                                                                            \
        if (a.rich_info && g1) hipLaunchKernelGGL((KERNEL<true, true>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (a.rich_info) hipLaunchKernelGGL((KERNEL<true, false>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (g1) hipLaunchKernelGGL((KERNEL<false, true>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, dim3(256), 0, s, __VA_ARGS__);                          \
    } while (0)

void ts_launch_render_fwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                          const ImageStateView &im, float *out_feature, float *out_depth, float *out_normal,
                          float *contrib_sum, float *contrib_max, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    TS_DISPATCH(render_fwd_kernel, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature, out_depth,
                out_normal, contrib_sum, contrib_max);
}

void ts_launch_render_bwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                          const ImageStateView &im, const float *dL_dout_feature, const float *dL_dout_depth,
                          const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    TS_DISPATCH(render_bwd_kernel, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, dL_dout_feature,
                dL_dout_depth, dL_dout_normal, grad_rec);
}
