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
//     their record gathers share the CU's L1, but they never synchronise (all LDS is wave-private, no barrier)
//     and each terminates as soon as its own 64 pixels are saturated;
//   * a batch = 64 list entries, one per lane: each lane gathers its entry's 64-byte record with four dwordx4
//     loads, does the per-(entry,quadrant) setup once (edge functions as affine forms of the in-quadrant pixel
//     offset; conservative support factor from alpha >= 1/255; bounding-box + separating-axis test against the
//     8x8 sample box) and the wave ballots the entries that can touch the quadrant (99.7 % of them then do);
//   * the wave then walks the set bits of that ballot (s_ff1 / s_flbit); per-entry constants reach all lanes as
//     LDS broadcasts (ds_read_b128 at a wave-uniform address), and the per-pixel test is 4 FMAs + min3 + 2 compares;
//   * forward: contrib_sum / contrib_max contributions are parked in LDS and reduced 8 entries at a time by two
//     transpose-reduce passes, so 8 entries cost one 8-lane atomic add and one 8-lane atomic max;
//   * backward: one scalar back-to-front composite per pixel instead of seven; every gradient term is a zeroth or
//     first pixel moment of three per-pair scalars, so the loop forms 19 products per lane, reduces them inside the
//     wave with transpose-reduce networks (v_permlane32_swap / v_permlane16_swap / DPP) that leave each sum in a
//     distinct lane quad, parks them in the entry's own (by then dead) LDS row, and once per batch the lane that
//     owns an entry turns its 19 sums into the 16 gradient values, flushed as one 16-lane global_atomic_add_f32
//     per 64-byte gradient record -- instead of the reference's 16 x 64 atomics per (tile quadrant, triangle).
//     An f32-MFMA formulation of the same sums exists (TS2D_BWD=mfma) and is ~10 % slower, see below.
//
// Skipping entries by the support box cannot change results: an entry is only skipped for a quadrant when no
// pixel of the quadrant can pass the reference's own tests (0 <= ecc <= 10 and alpha >= 1/255), and
// n_contrib / termination are tracked by list position exactly as the reference counts them.
#include "ts2d_common.h"
#include "ts2d_wave.h"

namespace
{
// Per-entry constants travel from the lane that owns the entry to all 64 pixel lanes through a wave-private LDS
// table read with uniform addresses (ds_read_b128 broadcast): that costs no VALU issue slots, whereas one
// v_readlane per constant costs ~4.3 cycles each (profiles/r01_valu_microbench.txt) in kernels that are VALU-bound.
// Row layout (CST floats per entry; 20-dword stride keeps the 8-lane ds_write_b128 groups conflict-free):
//   [0..3] A1 B1 C1 A2   [4..6] B2 C2 opacity   [8..11] r g b nx   [12..15] ny nz vd1 vd2   [16] vd3   [19] id
constexpr int CST = 20;

struct EntrySetup
{
    float A1, B1, C1, A2, B2, C2; // a1(q) = A1*qx + B1*qy + C1 (q = pixel offset inside the quadrant), same for a2
    float inv_area;               // 1 / area2
    float u1x, u1y, u2x, u2y, u3x, u3y; // screen vertices relative to the quadrant origin
    bool overlap;                 // the entry's support region meets the quadrant's 8x8 sample box
};

template <bool GAMMA1>
__device__ __forceinline__ EntrySetup entry_setup(float v1x, float v1y, float v2x, float v2y, float v3x, float v3y,
                                                  float op, float g2, float OX, float OY)
{
    EntrySetup s;
    const float area2 = (v2x - v1x) * (v3y - v1y) - (v2y - v1y) * (v3x - v1x); // the value the reference stores, forward.cu:137
    s.inv_area = 1.0f / area2;
    s.u1x = v1x - OX; s.u1y = v1y - OY; s.u2x = v2x - OX; s.u2y = v2y - OY; s.u3x = v3x - OX; s.u3y = v3y - OY;
    s.C1 = (s.u2x * s.u3y - s.u2y * s.u3x) * s.inv_area;
    s.A1 = (v2y - v3y) * s.inv_area;
    s.B1 = (v3x - v2x) * s.inv_area;
    s.C2 = (s.u3x * s.u1y - s.u3y * s.u1x) * s.inv_area;
    s.A2 = (v3y - v1y) * s.inv_area;
    s.B2 = (v1x - v3x) * s.inv_area;
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
    const float cx = (s.u1x + s.u2x + s.u3x) * (1.0f / 3.0f), cy = (s.u1y + s.u2y + s.u3y) * (1.0f / 3.0f);
    const float e1x = E * (s.u1x - cx), e2x = E * (s.u2x - cx), e3x = E * (s.u3x - cx);
    const float e1y = E * (s.u1y - cy), e2y = E * (s.u2y - cy), e3y = E * (s.u3y - cy);
    const float pad = 0.05f;
    const float bminx = cx + fminf(fminf(e1x, e2x), e3x) - pad, bmaxx = cx + fmaxf(fmaxf(e1x, e2x), e3x) + pad;
    const float bminy = cy + fminf(fminf(e1y, e2y), e3y) - pad, bmaxy = cy + fmaxf(fmaxf(e1y, e2y), e3y) + pad;
    // Separating-axis test of the E-scaled triangle against the quadrant's 8x8 sample box: box axes (the bbox
    // above) plus the three edge normals.  ecc <= E  <=>  min_i a_i >= (1 - E) / 3, and each a_i is affine in q,
    // so its maximum over the box is C_i + max(0, 7 A_i) + max(0, 7 B_i).
    // (A per-row interval coverage mask was tried instead: exact, but its ~200 VALU per batch cost more than the
    // few no-hit entries it removes -- see profiles/r01_notes.md.)
    const float m = (1.0f - E) * (1.0f / 3.0f);
    const float A3 = -s.A1 - s.A2, B3 = -s.B1 - s.B2, C3 = 1.0f - s.C1 - s.C2;
    const float max1 = s.C1 + fmaxf(0.0f, 7.0f * s.A1) + fmaxf(0.0f, 7.0f * s.B1);
    const float max2 = s.C2 + fmaxf(0.0f, 7.0f * s.A2) + fmaxf(0.0f, 7.0f * s.B2);
    const float max3 = C3 + fmaxf(0.0f, 7.0f * A3) + fmaxf(0.0f, 7.0f * B3);
    s.overlap = (E > 0.0f) && bminx <= 7.0f && bmaxx >= 0.0f && bminy <= 7.0f && bmaxy >= 0.0f && max1 >= m &&
                max2 >= m && max3 >= m;
    return s;
}

// Publishes the owning lane's constants for its entry.  Read back by every lane with a uniform row index.
template <bool RICH>
__device__ __forceinline__ void publish_entry(float *row, const EntrySetup &s, uint32_t id, const float4 &r1, const float4 &r2,
                                              const float4 &r3)
{
    float4 *q = (float4 *)row;
    q[0] = make_float4(s.A1, s.B1, s.C1, s.A2);
    q[1] = make_float4(s.B2, s.C2, r1.z, 0.0f);
    row[19] = __uint_as_float(id);
    q[2] = make_float4(r1.w, r2.x, r2.y, r2.z);
    if (RICH)
    {
        q[3] = make_float4(r2.w, r3.x, r3.y, r3.z);
        row[16] = r3.w;
    }
}

// Forward variant: exactly 16 floats per entry (the triangle id is not needed per pixel and travels by v_readlane):
//   [0..3] A1 B1 C1 A2   [4..7] B2 C2 opacity r   [8..11] g b nx ny   [12..15] nz vd1 vd2 vd3
constexpr int CSTF = 16;
template <bool RICH>
__device__ __forceinline__ void publish_entry_fwd(float *row, const EntrySetup &s, const float4 &r1, const float4 &r2,
                                                  const float4 &r3)
{
    float4 *q = (float4 *)row;
    q[0] = make_float4(s.A1, s.B1, s.C1, s.A2);
    q[1] = make_float4(s.B2, s.C2, r1.z, r1.w);
    q[2] = make_float4(r2.x, r2.y, r2.z, r2.w);
    if (RICH) q[3] = make_float4(r3.x, r3.y, r3.z, r3.w);
}

#ifdef TS2D_STATS
// Profiling builds only (-DTS2D_STATS): culling / occupancy statistics of render_fwd, read with ts2d_stats_read().
// [0] list entries visited  [1] entries surviving the setup cull  [2] entries with a stage-1 hit  [3] entries blended
// [4] (pixel, entry) pairs blended  [5] quadrants (waves)  [6] / [7] iterations if the four 4x4 blocks / two 8x4 halves of a
// quadrant walked their own blended entries in lockstep (sum over batches of the per-group maximum)
__device__ unsigned long long g_stats[8];
#define TS_STAT(i, v) st[i] += (unsigned long long)(v)
#else
#define TS_STAT(i, v)
#endif

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256) render_fwd_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                          const uint32_t *__restrict__ point_list,
                                                          const float4 *__restrict__ rec, float *__restrict__ final_T,
                                                          uint32_t *__restrict__ n_contrib, float *__restrict__ out_feature,
                                                          float *__restrict__ out_depth, float *__restrict__ out_normal,
                                                          float *__restrict__ contrib_sum, float *__restrict__ contrib_max)
{
    __shared__ __attribute__((aligned(16))) float cst_all[4][64 * CSTF];
    __shared__ float stage_all[RICH ? 4 : 1][8][64];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
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
    float *cst = cst_all[wave];
    float(*stage)[64] = stage_all[RICH ? wave : 0];

    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, anx = 0.0f, any_ = 0.0f, anz = 0.0f, ad = 0.0f;
    bool done = !inside;
    uint32_t last = (uint32_t)len; // a pixel that never saturates examines the whole list (forward.cu:296-297)

    // contrib_sum / contrib_max (forward.cu:323-324): the reference issues two global atomics per (pixel, triangle).
    // Here each contributing entry parks its 64 per-pixel contributions in a wave-private LDS slot; every 8
    // entries the 8 x 64 block is reduced by two transpose-reduce passes (sum, max) and leaves as ONE 8-lane
    // atomic add + ONE 8-lane atomic max.  (LDS float atomics are not an option: ds_add_f32 measures ~190
    // cycles per wave instruction on gfx950, see profiles/r01_lds_atomic_microbench.txt.)
    int staged = 0;          // entries parked so far (wave-uniform)
    uint32_t staged_ids = 0; // lane k holds the triangle id of parked entry k
    const int slot = RICH ? slot8_of_lane(lane) : 0;
    auto flush = [&]() {
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
            atomicMax((int *)contrib_max + gid, __float_as_int(rm)); // rm > 0: int order == float order
        }
        staged = 0;
    };

#ifdef TS2D_STATS
    unsigned long long st[8] = {0, 0, 0, 0, 0, 1, 0, 0};
#endif
    unsigned long long alive = __ballot(!done); // pixels that still blend (wave-uniform copy of !done)
    for (int base = 0; base < len; base += 64)
    {
        if (alive == 0) break;
        const int k = base + lane;
        const bool valid = k < len;
        TS_STAT(0, __popcll(__ballot(valid)));
        uint32_t id = 0;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (valid)
        {
            id = point_list[range.x + k] & TS_ID_MASK; // id bits (the top four are a quadrant mask, ts2d_support.h)
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            if (RICH) r3 = rp[3];
        }
        const EntrySetup s = entry_setup<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY);
        unsigned long long mask = __ballot(valid && s.overlap);
        TS_STAT(1, __popcll(mask));
        if (mask == 0) continue;
        publish_entry_fwd<RICH>(cst + lane * CSTF, s, r1, r2, r3);
#ifdef TS2D_STATS
        int cg4[4] = {0, 0, 0, 0}, cg2[2] = {0, 0};
#endif

        while (mask)
        {
            const int jc = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float4 c0 = *(const float4 *)(cst + jc * CSTF), c1 = *(const float4 *)(cst + jc * CSTF + 4);
            const float a1 = fmaf(c0.x, fx, fmaf(c0.y, fy, c0.z));
            const float a2 = fmaf(c0.w, fx, fmaf(c1.x, fy, c1.y));
            const float a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            bool hit = !done && ecc >= 0.0f && ecc <= 10.0f; // forward.cu:307
            if (__ballot(hit) == 0) continue;
            TS_STAT(2, 1);
            const float4 c2 = *(const float4 *)(cst + jc * CSTF + 8);
            float4 c3 = make_float4(0, 0, 0, 0);
            if (RICH) c3 = *(const float4 *)(cst + jc * CSTF + 12);
            const float pw = GAMMA1 ? ecc * ecc : pow_nonneg(ecc, g2);
            const float alpha = fminf(0.99f, c1.z * fast_exp(-0.5f * pw)); // forward.cu:311-312
            hit = hit && alpha >= 1.0f / 255.0f;                           // forward.cu:313
            if (__ballot(hit) == 0) continue;
            TS_STAT(3, 1);
            TS_STAT(4, __popcll(__ballot(hit)));
#ifdef TS2D_STATS
            {
                const unsigned long long hb = __ballot(hit);
                // lanes are ly * 8 + lx: 4x4 blocks and 8x4 halves of the quadrant
                if (hb & 0x000000000F0F0F0Full) cg4[0]++;
                if (hb & 0x00000000F0F0F0F0ull) cg4[1]++;
                if (hb & 0x0F0F0F0F00000000ull) cg4[2]++;
                if (hb & 0xF0F0F0F000000000ull) cg4[3]++;
                if (hb & 0x00000000FFFFFFFFull) cg2[0]++;
                if (hb & 0xFFFFFFFF00000000ull) cg2[1]++;
            }
#endif
            // Branch-free blend: lanes that do not hit run with alpha = 0, which leaves every accumulator and T
            // bit-unchanged (x + c*0 == x, T*1 == T).
            const float al = hit ? alpha : 0.0f;
            const float contrib = al * T;
            ar = fmaf(c1.w, contrib, ar);
            ag = fmaf(c2.x, contrib, ag);
            ab = fmaf(c2.y, contrib, ab);
            if (RICH)
            {
                anx = fmaf(c2.z, contrib, anx);
                any_ = fmaf(c2.w, contrib, any_);
                anz = fmaf(c3.x, contrib, anz);
                const float d = c3.y * a1 + c3.z * a2 + c3.w * a3; // forward.cu:328
                ad = fmaf(d, contrib, ad);
                stage[staged][lane] = contrib;
                staged_ids = (lane == staged) ? bcast(id, jc) : staged_ids;
                if (++staged == 8) flush();
            }
            T *= (1.0f - al);
            if (hit && T <= 0.0001f) // forward.cu:333
            {
                done = true;
                last = (uint32_t)(base + jc + 1);
            }
            alive = __ballot(!done);
            if (alive == 0)
            {
                mask = 0;
                base = len; // leave both loops
            }
        }
#ifdef TS2D_STATS
        st[6] += (unsigned long long)max(max(cg4[0], cg4[1]), max(cg4[2], cg4[3]));
        st[7] += (unsigned long long)max(cg2[0], cg2[1]);
#endif
    }
    if (RICH && staged > 0) flush();
#ifdef TS2D_STATS
    if (lane == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_stats[i], st[i]);
#endif

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
            out_depth[pix] = ad + T * (a.background_depth_dev ? *a.background_depth_dev : a.background_depth); // forward.cu:349
            out_normal[pix] = anx;
            out_normal[HW + pix] = any_;
            out_normal[2 * HW + pix] = anz;
        }
    }
}

// Backward.  Per (pixel, triangle) pair the reference adds 16 values into per-triangle arrays (backward.cu:412-490).
// All of them are linear in three per-pair scalars with per-PIXEL weights:
//     contrib             -> dL/drgb (weights dL_dpix_rgb), dL/dnormal (dL_dpix_normal), w = dL_dpix_depth * contrib
//     z = -3 dL/decc      -> enters dL/da_k of the arg-min barycentric k only
//     dL/dalpha * G       -> dL/dopacity
// and the six screen-space vertex gradients and three v_depth gradients are per-TRIANGLE linear maps of the
// zeroth and first pixel moments of w and z_k (a_k and p_v_k = u_k - q are affine in the pixel offset q).  The hot
// loop therefore only forms 19 products per lane (moments + colour/normal/opacity terms), reduces them across the
// wave, parks the 19 sums of entry j in LDS, and once per batch the lane that OWNS entry j turns them into the 16
// gradient values with its own (register-resident) triangle constants and issues the global atomics.
// MFMA = true (experimental, env TS2D_BWD=mfma): the 19 per-entry sums are formed on the matrix cores instead of by
// cross-lane VALU reductions.  Measured on MI355X it is ~10 % SLOWER than the VALU path (render_bwd 1.73 vs 1.56 ms,
// profiles/r01_notes.md): f32 MFMA runs at the f32 vector rate, the padded 16 x 16 x 64 product per 3 entries costs as
// many matrix cycles (171 per entry) as the reduction network costs VALU cycles, and the LDS tile lowers occupancy from
// 7 to 4 waves per SIMD.  Kept as the worked answer to "can MFMA help here?" and as a base for a split-f16 variant.  Every sum has the form sum_pixels f(pixel) * g(entry, pixel) with f one of 12 per-pixel constants
// {1, qx, qy, dL/dpix of r g b nx ny nz, dL/ddepth * {1, qx, qy}} and g one of 5 per-entry fields {contrib, z1, z2, z3,
// dL_dalpha * G}: a (rows = entry fields) x (K = 64 pixels) x (cols = per-pixel constants) matrix product.  Each
// processed entry parks its 5 fields as rows of a wave-private LDS tile (the transposition: a field is computed one
// pixel per lane, the MFMA wants one ROW per lane); every 3 entries (15 rows) sixteen v_mfma_f32_16x16x4_f32 contract
// the tile with the constant operand held in 16 VGPRs, and the lanes that hold useful elements of D scatter them into
// the entries' sum slots.  f32 MFMA is exact f32 (an fmaf chain) and replaces 15 permlane swaps + 15 adds + 9 DPP ops
// + 12 multiplies per entry with 5 ds_write_b32, 4/3 ds_read_b128 and 16/3 MFMA issues.
constexpr int MROWS = 16, MRS = 68; // LDS tile: 16 rows (3 entries x 5 fields + 1 spare), row stride 68 floats (16-B aligned, bank-staggered)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: lane l supplies A[m = l & 15][k = l >> 4] and B[k = l >> 4][n = l & 15]; chunk c of the 16
// contracts the pixels {16 k + c}, so that a lane's A elements of all chunks are 16 consecutive floats of its row.
// D[4 (l >> 4) + i][l & 15] sits in register i of lane l.  Everything is passed by value so that it stays in registers
// (a by-reference lambda capture sent the constant operand and the counters to scratch memory).
__device__ __forceinline__ void mfma_reduce_set(const float *mt, float *sums, int lane, float4 B0, float4 B1, float4 B2, float4 B3,
                                                int t0, int t1, int t2, int t3, int nslot, int jc0, int jc1, int jc2)
{
    const float *arow = mt + (lane & 15) * MRS + 16 * (lane >> 4);
    const float4 q0 = *(const float4 *)(arow), q1 = *(const float4 *)(arow + 4), q2 = *(const float4 *)(arow + 8),
                 q3 = *(const float4 *)(arow + 12);
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q0.x, B0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q0.y, B0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q0.z, B0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q0.w, B0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q1.x, B1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q1.y, B1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q1.z, B1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q1.w, B1.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q2.x, B2.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q2.y, B2.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q2.z, B2.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q2.w, B2.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q3.x, B3.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q3.y, B3.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q3.z, B3.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q3.w, B3.w, acc, 0, 0, 0);
    const int t[4] = {t0, t1, t2, t3};
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int sl = t[i] >> 8;
        if (t[i] >= 0 && sl < nslot)
        {
            const int jcs = sl == 0 ? jc0 : (sl == 1 ? jc1 : jc2);
            sums[jcs * CST + (t[i] & 0xFF)] = acc[i];
        }
    }
}

template <bool RICH, bool GAMMA1, bool MFMA>
__global__ void __launch_bounds__(256, MFMA ? 4 : 7) render_bwd_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                          const uint32_t *__restrict__ point_list,
                                                          const float4 *__restrict__ rec, const float *__restrict__ final_T,
                                                          const uint32_t *__restrict__ n_contrib,
                                                          const float *__restrict__ dL_dout_feature,
                                                          const float *__restrict__ dL_dout_depth,
                                                          const float *__restrict__ dL_dout_normal, float *__restrict__ grad_rec)
{
    __shared__ __attribute__((aligned(16))) float cst_all[4][64 * CST];
    __shared__ __attribute__((aligned(16))) float tile_all[MFMA ? 4 : 1][MFMA ? MROWS * MRS : 4];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
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
    float *cst = cst_all[wave];
    float *mt = tile_all[MFMA ? wave : 0];
    // The 19 reduced sums of an entry are parked in the entry's own constants row: the row is dead once the entry has
    // been processed (each entry is visited once per batch; LDS executes a wave's accesses in order) except for the
    // triangle id in slot 19.
    float *sums = cst;

    float T = inside ? final_T[pix] : 0.0f;            // backward.cu:318
    const int last = inside ? (int)n_contrib[pix] : 0; // backward.cu:320
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
            B = fmaf(dd, a.background_depth_dev ? *a.background_depth_dev : a.background_depth, B); // accum_normal starts at 0, accum_depth at background_depth
        }
    }
    const int slot = slot_of_lane(lane);
    const int slot4 = RICH ? slot4_of_lane(lane) : 0;
    const bool writer16 = (lane & 3) == 0, writer4 = RICH && (lane & 15) == 0;

    // entries at list positions >= max(last) are skipped by every pixel of the quadrant (backward.cu:377-379)
    const int wlast = __builtin_amdgcn_readlane(__float_as_int(wave_max63_nonneg((float)last)), 63);
    const int maxlast = (int)__int_as_float(wlast);
    if (maxlast <= 0) return;

    // ---- MFMA path: constant operand (16 VGPRs) and scatter targets (see mfma_reduce_set) -----------------------------
    float4 Bq0 = make_float4(0, 0, 0, 0), Bq1 = Bq0, Bq2 = Bq0, Bq3 = Bq0;
    int tg0 = -1, tg1 = -1, tg2 = -1, tg3 = -1; // per D register: (slot << 8) | sum index, or -1 when the element is not one of the 19 sums
    if (MFMA)
    {
        float *tmp = mt; // [pixel][8]: the per-pixel upstream gradients, pixel-major
        tmp[lane * 8 + 0] = dpr; tmp[lane * 8 + 1] = dpg; tmp[lane * 8 + 2] = dpb;
        tmp[lane * 8 + 3] = dnx; tmp[lane * 8 + 4] = dny; tmp[lane * 8 + 5] = dnz; tmp[lane * 8 + 6] = dd;
        const int n = lane & 15, kq = lane >> 4;
        const int src = (n >= 3 && n <= 8) ? n - 3 : 6;
        auto bval = [=](int c) -> float {
            const int p = 16 * kq + c;
            const float qx = (float)(p & 7), qy = (float)(p >> 3);
            const float t = tmp[p * 8 + src];
            float val = 0.0f;
            if (n == 0) val = 1.0f;
            else if (n == 1) val = qx;
            else if (n == 2) val = qy;
            else if (n <= 9) val = t;       // dL/d(r g b nx ny nz), dL/ddepth
            else if (n == 10) val = t * qx; // dL/ddepth * qx
            else if (n == 11) val = t * qy;
            return val;
        };
        Bq0 = make_float4(bval(0), bval(1), bval(2), bval(3));
        Bq1 = make_float4(bval(4), bval(5), bval(6), bval(7));
        Bq2 = make_float4(bval(8), bval(9), bval(10), bval(11));
        Bq3 = make_float4(bval(12), bval(13), bval(14), bval(15));
        auto target = [=](int i) -> int {
            const int R = 4 * kq + i, sl = R / 5, vec = R % 5;
            int idx = -1;
            if (R < 15)
            {
                if (vec == 0 && n >= 3 && n <= 11) idx = 7 + n;                  // rgb 10..12, normal 13..15, depth moments 16..18
                else if (vec >= 1 && vec <= 3 && n < 3) idx = 3 * (vec - 1) + n; // z_k moments 0..8
                else if (vec == 4 && n == 0) idx = 9;                            // dL/dopacity
            }
            return idx < 0 ? -1 : ((sl << 8) | idx);
        };
        tg0 = target(0); tg1 = target(1); tg2 = target(2); tg3 = target(3);
    }
    int nslot = 0, jc0 = 0, jc1 = 0, jc2 = 0; // entries parked in the tile (wave-uniform)

    for (int base = ((maxlast - 1) >> 6) << 6; base >= 0; base -= 64)
    {
        const int k = base + lane;
        const bool valid = k < maxlast;
        uint32_t id = 0;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (valid)
        {
            id = point_list[range.x + k] & TS_ID_MASK; // id bits (the top four are a quadrant mask, ts2d_support.h)
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            if (RICH) r3 = rp[3];
        }
        const EntrySetup s = entry_setup<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY);
        unsigned long long mask = __ballot(valid && s.overlap);
        if (mask == 0) continue;
        publish_entry<RICH>(cst + lane * CST, s, id, r1, r2, r3);
        unsigned long long touched = 0; // entries of this batch that received gradient sums

        while (mask)
        {
            const int jc = 63 - __builtin_clzll(mask);
            mask &= ~(1ull << jc);
            const float4 c0 = *(const float4 *)(cst + jc * CST), c1 = *(const float4 *)(cst + jc * CST + 4);
            const float a1 = fmaf(c0.x, fx, fmaf(c0.y, fy, c0.z));
            const float a2 = fmaf(c0.w, fx, fmaf(c1.x, fy, c1.y));
            const float a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            bool hit = (base + jc < last) && ecc >= 0.0f && ecc <= 10.0f; // backward.cu:378,393
            if (__ballot(hit) == 0) continue;
#ifdef TS2D_ABLATION
            if (a.ablate == 3) { T += ecc * 1e-30f; continue; } // profiling: stage 1 only
#endif
            const float4 c2 = *(const float4 *)(cst + jc * CST + 8);
            float4 c3 = make_float4(0, 0, 0, 0);
            float c4 = 0.0f;
            if (RICH)
            {
                c3 = *(const float4 *)(cst + jc * CST + 12);
                c4 = cst[jc * CST + 16];
            }
            const float pw = GAMMA1 ? ecc * ecc : pow_nonneg(ecc, g2);
            const float power = -0.5f * pw;
            const float op = c1.z;
            const float G = fast_exp(power);
            const float alpha = fminf(0.99f, op * G);
            hit = hit && alpha >= 1.0f / 255.0f; // backward.cu:400
            if (__ballot(hit) == 0) continue;
#ifdef TS2D_ABLATION
            if (a.ablate == 2) { T = T * __builtin_amdgcn_rcpf(1.0f - (hit ? alpha : 0.0f)); continue; } // profiling
#endif

            // Branch-free from here on: lanes that do not hit run with alpha = 0 so that T, B stay bit-unchanged
            // and every term they produce is an exact 0 (all terms carry a factor alpha, contrib or `hit`).
            const float al = hit ? alpha : 0.0f;
            const float oma = 1.0f - al;
            T = T * __builtin_amdgcn_rcpf(oma); // backward.cu:403
            const float contrib = al * T;
            float X = fmaf(dpb, c2.z, fmaf(dpg, c2.y, dpr * c2.x)); // backward.cu:415
            float w = 0.0f;
            if (RICH) // backward.cu:419-437
            {
                X = fmaf(dnz, c3.y, fmaf(dny, c3.x, fmaf(dnx, c2.w, X)));
                const float depth = fmaf(c4, a3, fmaf(c3.w, a2, c3.z * a1));
                X = fmaf(dd, depth, X);
                w = dd * contrib; // dL_ddepth
            }
            const float dL_dcontrib = X - B;
            B = fmaf(al, X, oma * B);
            const float dL_dalpha = dL_dcontrib * T;
            // backward.cu:443-447: dL_decc = dL_dpower * 2 gamma * power / (ecc + 1e-8), dL_dpower = dL_dalpha * alpha
            // unless the 0.99 clamp was active.  The select sits last so that a non-hit lane never multiplies 0 * inf.
            const float decc_raw = dL_dalpha * alpha * g2 * power * __builtin_amdgcn_rcpf(ecc + 1e-8f);
            const float z = (hit && op * G < 0.99f) ? -3.0f * decc_raw : 0.0f;
            const bool k1 = a1 <= a2 && a1 <= a3;        // backward.cu:449-461 (ties: a1, then a2)
            const bool k2 = !k1 && a2 <= a1 && a2 <= a3;
            const float z1 = k1 ? z : 0.0f, z2 = k2 ? z : 0.0f, z3 = (k1 || k2) ? 0.0f : z;

            if (MFMA)
            {
                float *row = mt + (nslot * 5) * MRS + lane; // this pixel's column of the entry's five rows
                row[0] = contrib;
                row[MRS] = z1; row[2 * MRS] = z2; row[3 * MRS] = z3;
                row[4 * MRS] = hit ? dL_dalpha * G : 0.0f; // backward.cu:490 (not gated by the clamp)
                if (nslot == 0) jc0 = jc; else if (nslot == 1) jc1 = jc; else jc2 = jc;
                touched |= 1ull << jc;
                if (++nslot == 3)
                {
                    mfma_reduce_set(mt, sums, lane, Bq0, Bq1, Bq2, Bq3, tg0, tg1, tg2, tg3, nslot, jc0, jc1, jc2);
                    nslot = 0;
                }
                continue;
            }
            float v[16];
            v[0] = z1; v[1] = z1 * fx; v[2] = z1 * fy;
            v[3] = z2; v[4] = z2 * fx; v[5] = z2 * fy;
            v[6] = z3; v[7] = z3 * fx; v[8] = z3 * fy;
            v[9] = hit ? dL_dalpha * G : 0.0f; // backward.cu:490 (not gated by the clamp)
            v[10] = dpr * contrib; v[11] = dpg * contrib; v[12] = dpb * contrib; // backward.cu:412
            v[13] = dnx * contrib; v[14] = dny * contrib; v[15] = dnz * contrib; // backward.cu:421-423
#ifdef TS2D_ABLATION
            if (a.ablate == 1) // profiling: everything except the cross-lane reductions
            {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("" ::"v"(v[i]));
                asm volatile("" ::"v"(w));
                continue;
            }
#endif
            const float r16 = reduce16(v, lane);
            if (writer16) sums[jc * CST + slot] = r16;
            if (RICH)
            {
                const float r4 = reduce4(w, w * fx, w * fy, 0.0f);
                if (writer4 && slot4 < 3) sums[jc * CST + 16 + slot4] = r4; // slot 19 keeps the triangle id
            }
            touched |= 1ull << jc;
        }

        // Batch epilogue: the lane that owns a touched entry converts the 19 sums into the 16 gradient values.
        if (touched == 0) continue;
        if (MFMA && nslot > 0) // entries still parked in the tile
        {
            mfma_reduce_set(mt, sums, lane, Bq0, Bq1, Bq2, Bq3, tg0, tg1, tg2, tg3, nslot, jc0, jc1, jc2);
            nslot = 0;
        }
        if ((touched >> lane) & 1)
        {
            const float4 *sq = (const float4 *)(sums + lane * CST);
            const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
            float W0 = 0.0f, Wx = 0.0f, Wy = 0.0f;
            if (RICH)
            {
                const float4 s4 = sq[4];
                W0 = s4.x; Wx = s4.y; Wy = s4.z;
            }
            const float vd1 = r3.y, vd2 = r3.z, vd3 = r3.w;
            // zeroth / first moments of dL/da_k = w * vd_k + z_k   (backward.cu:433,462)
            const float D10 = fmaf(vd1, W0, s0.x), D1x = fmaf(vd1, Wx, s0.y), D1y = fmaf(vd1, Wy, s0.z);
            const float D20 = fmaf(vd2, W0, s0.w), D2x = fmaf(vd2, Wx, s1.x), D2y = fmaf(vd2, Wy, s1.y);
            const float D30 = fmaf(vd3, W0, s1.z), D3x = fmaf(vd3, Wx, s1.w), D3y = fmaf(vd3, Wy, s2.x);
            const float A3 = -s.A1 - s.A2, B3 = -s.B1 - s.B2, C3 = 1.0f - s.C1 - s.C2;
            // S = sum_pixels sum_k dL/da_k * a_k with a_k = A_k qx + B_k qy + C_k
            const float S = s.A1 * D1x + s.B1 * D1y + s.C1 * D10 + s.A2 * D2x + s.B2 * D2y + s.C2 * D20 + A3 * D3x + B3 * D3y +
                            C3 * D30;
            // backward.cu:464-479 regrouped: with E_k = perp(opposite edge of vertex k) / area2 = -(A_k, B_k):
            //   dL/dv1 = S*E1 + perp(sum(da3*p_v2 - da2*p_v3))/area2, cyclically; p_v_k = u_k - q.
            const float t1x = s.u2x * D30 - D3x - s.u3x * D20 + D2x, t1y = s.u2y * D30 - D3y - s.u3y * D20 + D2y;
            const float t2x = s.u3x * D10 - D1x - s.u1x * D30 + D3x, t2y = s.u3y * D10 - D1y - s.u1y * D30 + D3y;
            const float t3x = s.u1x * D20 - D2x - s.u2x * D10 + D1x, t3y = s.u1y * D20 - D2y - s.u2y * D10 + D1y;
            const float ia = s.inv_area;
            // park the 16 gradient values in the entry's own LDS row (it has just been read), in grad-record order
            float4 *gq = (float4 *)(sums + lane * CST);
            gq[0] = make_float4(ia * t1y - S * s.A1, -ia * t1x - S * s.B1, ia * t2y - S * s.A2, -ia * t2x - S * s.B2);
            gq[1] = make_float4(ia * t3y - S * A3, -ia * t3x - S * B3, s2.y /* dL/dopacity */, s2.z /* dL/drgb */);
            gq[2] = make_float4(s2.w, s3.x, s3.y /* dL/dnormal_view */, s3.z);
            // dL/dv_depth_k = sum w * a_k   (backward.cu:429-431)
            gq[3] = make_float4(s3.w, s.A1 * Wx + s.B1 * Wy + s.C1 * W0, s.A2 * Wx + s.B2 * Wy + s.C2 * W0,
                                A3 * Wx + B3 * Wy + C3 * W0);
        }
        // Coalesced flush: 16 consecutive lanes add the 16 floats (one 64-byte line) of one triangle's gradient
        // record, four touched-or-not entries per instruction.  (One atomic per lane-and-value instead would be
        // 16x the atomic requests: measured 4.6 ms for this scene.)
        {
            const int sub = lane >> 4, col = lane & 15;
#pragma unroll 1
            for (int e0 = 0; e0 < 64; e0 += 4)
            {
                if (((touched >> e0) & 0xFull) == 0) continue;
                const int e = e0 + sub;
                if ((touched >> e) & 1)
                {
                    const uint32_t eid = __float_as_uint(cst[e * CST + 19]);
#ifdef TS2D_ABLATION
                    if (a.ablate == 4) continue; // profiling: no gradient-record atomics
#endif
                    if (RICH || col < 10) unsafeAtomicAdd(grad_rec + TS_GRAD_FLOATS * (size_t)eid + col, sums[e * CST + col]);
                }
            }
        }
    }
}
} // namespace

#define TS_DISPATCH_BWD(MFMA, ...)                                                                                    \
    do                                                                                                                \
    {                                                                                                                 \
        const bool g1 = (a.gamma == 1.0f);                                                                            \
        if (a.rich_info && g1) hipLaunchKernelGGL((render_bwd_kernel<true, true, MFMA>), grid, dim3(256), 0, s, __VA_ARGS__);   \
        else if (a.rich_info) hipLaunchKernelGGL((render_bwd_kernel<true, false, MFMA>), grid, dim3(256), 0, s, __VA_ARGS__);   \
        else if (g1) hipLaunchKernelGGL((render_bwd_kernel<false, true, MFMA>), grid, dim3(256), 0, s, __VA_ARGS__);            \
        else hipLaunchKernelGGL((render_bwd_kernel<false, false, MFMA>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
    } while (0)

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
    const dim3 grid((unsigned)(a.grid_x * a.grid_y ? ts_tile_units(a.grid_x, a.grid_y) : 0));
    if (grid.x == 0) return;
    TS_DISPATCH(render_fwd_kernel, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature, out_depth,
                out_normal, contrib_sum, contrib_max);
}

void ts_launch_render_bwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                          const ImageStateView &im, const float *dL_dout_feature, const float *dL_dout_depth,
                          const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y ? ts_tile_units(a.grid_x, a.grid_y) : 0));
    if (grid.x == 0) return;
    if (!a.bwd_mfma) // default: cross-lane (permlane / DPP) reduction networks; TS2D_BWD=mfma selects the matrix-core variant
        TS_DISPATCH_BWD(false, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, dL_dout_feature, dL_dout_depth,
                        dL_dout_normal, grad_rec);
    else
        TS_DISPATCH_BWD(true, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, dL_dout_feature, dL_dout_depth,
                        dL_dout_normal, grad_rec);
}

#ifdef TS2D_STATS
extern "C" __attribute__((visibility("default"))) int ts2d_stats_read(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * 8);
    if (e == hipSuccess && reset)
    {
        unsigned long long z[8] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : 2;
}
#endif
