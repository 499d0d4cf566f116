// render_group.hip -- per-pixel alpha blend (forward) and its back-to-front replay (backward), lane-group edition.
//
// Behaviour follows FORWARD::renderCUDA (R2D/src/forward.cu:198-355) and BACKWARD::renderCUDA
// (R2D/src/backward.cu:265-493); SURVEY.md Appendix B lists the quirks that are kept (integer pixel centres, n_contrib
// counts examined entries, stop AFTER the triangle that drives T <= 1e-4, dL_dopacity not gated by the 0.99 clamp,
// arg-min tie order a1, a2, a3, division by ecc + 1e-8).
//
// Why this structure (measured on MI355X, profiles/r02_notes.md):
//   * a blended (triangle, 8x8 quadrant) pair touches 18 of the 64 pixels on average, so one triangle per wave
//     iteration leaves 72 % of the lanes idle, and gfx950 does not skip an all-idle 32-lane pass;
//   * on gfx950 only fma/add/mul (f32) and add/and (u32) issue at the full 32-lanes-per-clock rate; v_cmp, v_cndmask,
//     v_min/max, every DPP form and the integer shift/mad forms are half rate, transcendentals and v_permlane*_swap
//     quarter rate -- the blend loops are bound by exactly those, not by FMAs.
// So: one wave64 still owns one 8x8 pixel quadrant of a 16x16 tile, but its lanes form FOUR 16-lane groups, one per 4x4
// pixel block, and every group walks ITS OWN culled list of the batch's triangles: four different triangles are blended
// per wave step (lane occupancy 28 % -> ~45 %), the per-step body is branch-free, and all cross-lane reductions stay
// inside a 16-lane DPP row (no v_permlane*_swap).
//
//   batch   = 64 list entries, one per lane: the lane gathers the 64-byte render record, computes the conservative
//             support of the triangle (edge functions as affine forms of the in-quadrant pixel offset, used for CULLING
//             only) against the four 4x4 blocks, and the wave ballots one 64-bit mask per block;
//   lists   = each block's surviving entries, compacted in visiting order into a 64-byte LDS list (v_mbcnt rank);
//   step    = every lane reads ITS group's next entry index, then that entry's constants from the wave-private LDS
//             table (4 distinct rows per ds_read_b128 cost the same as one broadcast row, tools/valu_bench2.hip); a group
//             whose list is exhausted reads the dummy row -1, which no pixel can hit;
//   pixels  = barycentrics are evaluated exactly as the reference does, cross(v_j - p, v_k - p) / area2 from
//             pixel-relative vertex offsets (v - tile origin and (v - origin) - offset are exact in fp32, so the offsets
//             are bit-identical to the reference's).  Round 1 used affine forms of the pixel offset instead: 4 FMAs
//             cheaper, but ~10x noisier on sub-pixel slivers, where the 1/area2 amplification turns 1e-6 into 1e-3
//             (profiles/r02_noise_floor_1M_before.json) -- the gradients of those few triangles dominate the norm.
#include "ts2d_common.h"
#include "ts2d_wave.h"
#include "ts2d_group.h"
#include "ts2d_support.h"

#ifndef TSG_FWD_WAVES // resident waves per SIMD the register budget is declared for (occupancy experiments: tools/build_variant.sh ... -DTSG_BWD_WAVES=8)
#define TSG_FWD_WAVES 7
#endif
#ifndef TSG_BWD_WAVES
#define TSG_BWD_WAVES 7
#endif
#ifndef TSG_FWD_CAP // entries per dense batch (ts2d_group.h: stream_refill)
#define TSG_FWD_CAP 64
#endif
#ifndef TSG_BWD_CAP
#define TSG_BWD_CAP 64
#endif
#ifndef TSG_TCAP
#define TSG_TCAP 960
#endif
#ifndef TSG_PART // which kernels this translation unit holds: 1 = forward, 2 = backward, 3 = both.  build.py compiles the file twice, so that each
#define TSG_PART 3 // kernel gets its own scheduler strategy (round 6: max-ilp is +1.3 % for the forward and -1 % for the backward, profiles/r05_notes.md)
#endif
#ifndef TSG_CARRY // 1: a batch with more than NR surviving entries hands the ones beyond the table to the NEXT batch instead of taking a second pass
#define TSG_CARRY 0 // ("carried-over table", DESIGN 14 / VERDICT r5 item 3 (i): group_sim -26 % passes.  Built and measured in round 6, parity-green:
#endif              // render_fwd 0.382 -> 0.403 ms, render_bwd 0.792 -> 0.806 -- the carried entries take slots of the next batch, i.e. MORE batches, each
                    // with its full cull and list build, where a second pass had neither; profiles/r06_blend_ab.txt)
#ifndef TSG_PROBE
#define TSG_PROBE 0 // profiling builds: 1 = no contribution atomics, 2 = no contribution statistics at all, 3 = no serialised accumulate,
                    // 4 = backward without its step loop (what the per-batch work alone costs), 6 = backward without the row flush,
                    // 7 = backward whose row flush is a plain store instead of an atomic add -- results wrong
#endif
namespace
{
// ROW (ts2d_group.h) = 20 floats per entry row of the constants table:
//   [0..3] u1x u1y u2x u2y   [4..7] u3x u3y 1/area2 opacity   [8..11] r g b nx   [12..15] ny nz vd1 vd2   [16] vd3   [17] id   [18] the entry's position in the tile's list
// (u_k = screen vertex k relative to the quadrant origin); row -1 is a dummy that fails every pixel's ecc test.  The backward appends the
// entry's 16 gradient sums to the row (BROW floats).  A list entry is the LDS BYTE OFFSET of its row (u16): the step loops spend no
// instruction on unpacking or scaling an index (round 3: four half-rate instructions per step gone; gfx950 issues shifts, bit-field
// extracts and 24-bit multiply-adds at half rate, tools/valu_bench3.hip).
[[maybe_unused]] constexpr int BROW = ROW + 16;

struct BlockCull
{
    float u1x, u1y, u2x, u2y, u3x, u3y, ia;
    bool ov[4]; // the triangle's support (alpha >= 1/255 and ecc <= 10) can reach block g = (by >> 2) * 2 + (bx >> 2)
};

__device__ __forceinline__ void publish_row(float *row, const BlockCull &s, uint32_t id, int jpos, const float4 &r1, const float4 &r2, const float4 &r3)
{
    float4 *q = (float4 *)row;
    q[0] = make_float4(s.u1x, s.u1y, s.u2x, s.u2y);
    q[1] = make_float4(s.u3x, s.u3y, s.ia, r1.z);
    q[2] = make_float4(r1.w, r2.x, r2.y, r2.z);
    q[3] = make_float4(r2.w, r3.x, r3.y, r3.z);
    q[4] = make_float4(r3.w, __uint_as_float(id), __int_as_float(jpos), 0.0f);
}

// The part of the setup that ends up in the entry's table row: vertices relative to the quadrant origin, 1 / area2.
__device__ __forceinline__ void entry_geometry(BlockCull &s, float v1x, float v1y, float v2x, float v2y, float v3x, float v3y, float OX, float OY)
{
    // area2 exactly as preprocess evaluates (and the reference stores) it: cross(v2 - v1, v3 - v1) without contraction
    const float area2 = __fsub_rn(__fmul_rn(v2x - v1x, v3y - v1y), __fmul_rn(v2y - v1y, v3x - v1x)); // forward.cu:137
    s.ia = __builtin_amdgcn_rcpf(area2); // the reference divides by area2 per pixel; a 1-ulp reciprocal moves a_k by <= 2 ulp
    s.u1x = v1x - OX; s.u1y = v1y - OY; s.u2x = v2x - OX; s.u2y = v2y - OY; s.u3x = v3x - OX; s.u3y = v3y - OY;
}

// Second pass of a batch with more than NR surviving entries (rare): the lane gathers its entry's record again -- keeping the
// first gather's registers alive across the first pass would cost the occupancy the compaction buys.
template <bool RICH>
__device__ __forceinline__ uint32_t republish_row(float *row, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, uint32_t pos,
                                                  int jpos, float OX, float OY)
{
    const uint32_t id = point_list[pos] & TS_ID_MASK; // the top bits are the instance's quadrant mask
    const float4 *rp = rec + 4 * (size_t)id;
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = RICH ? rp[3] : make_float4(0, 0, 0, 0);
    BlockCull s;
    entry_geometry(s, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, OX, OY);
    publish_row(row, s, id, jpos, r1, r2, r3);
    return id;
}

// Conservative culling of one triangle against the four 4x4 sample blocks of the quadrant whose origin is (OX, OY).
// The affine forms a_k(q) = A_k qx + B_k qy + C_k are used ONLY here; their rounding error (up to ~(|C| + 7|A| + 7|B|) ulp
// for sub-pixel slivers) is added to the acceptance margin so that no pixel the exact test would blend is ever culled.
template <bool GAMMA1>
__device__ __forceinline__ BlockCull block_cull(float v1x, float v1y, float v2x, float v2y, float v3x, float v3y, float op, float g2,
                                                float OX, float OY)
{
    BlockCull s;
    entry_geometry(s, v1x, v1y, v2x, v2y, v3x, v3y, OX, OY);
    const float C1 = (s.u2x * s.u3y - s.u2y * s.u3x) * s.ia, A1 = (v2y - v3y) * s.ia, B1 = (v3x - v2x) * s.ia;
    const float C2 = (s.u3x * s.u1y - s.u3y * s.u1x) * s.ia, A2 = (v3y - v1y) * s.ia, B2 = (v1x - v3x) * s.ia;
    const float A3 = -A1 - A2, B3 = -B1 - B2, C3 = 1.0f - C1 - C2;
    // alpha >= 1/255 needs ecc^(2 gamma) <= 2 ln(255 op); ecc <= E is the triangle scaled by E about its centroid
    const float E = support_scale<GAMMA1>(op, g2); // ts2d_support.h
    const float cx = (s.u1x + s.u2x + s.u3x) * (1.0f / 3.0f), cy = (s.u1y + s.u2y + s.u3y) * (1.0f / 3.0f);
    const float e1x = E * (s.u1x - cx), e2x = E * (s.u2x - cx), e3x = E * (s.u3x - cx);
    const float e1y = E * (s.u1y - cy), e2y = E * (s.u2y - cy), e3y = E * (s.u3y - cy);
    const float pad = 0.05f;
    const float bminx = cx + fminf(fminf(e1x, e2x), e3x) - pad, bmaxx = cx + fmaxf(fmaxf(e1x, e2x), e3x) + pad;
    const float bminy = cy + fminf(fminf(e1y, e2y), e3y) - pad, bmaxy = cy + fmaxf(fmaxf(e1y, e2y), e3y) + pad;
    const bool live = E > 0.0f;
    const bool x0 = live && bminx <= 3.0f && bmaxx >= 0.0f, x1 = live && bminx <= 7.0f && bmaxx >= 4.0f;
    const bool y0 = bminy <= 3.0f && bmaxy >= 0.0f, y1 = bminy <= 7.0f && bmaxy >= 4.0f;
    // separating axes = the three edge normals: ecc <= E  <=>  min_k a_k >= (1 - E) / 3, and the maximum of a_k over the
    // 4x4 sample box at (bx, by) is C_k + A_k bx + B_k by + max(0, 3 A_k) + max(0, 3 B_k)
    const float m = (1.0f - E) * (1.0f / 3.0f);
    const float k1 = C1 + fmaxf(0.0f, 3.0f * A1) + fmaxf(0.0f, 3.0f * B1) - m + 1e-6f * (fabsf(C1) + 7.0f * (fabsf(A1) + fabsf(B1)));
    const float k2 = C2 + fmaxf(0.0f, 3.0f * A2) + fmaxf(0.0f, 3.0f * B2) - m + 1e-6f * (fabsf(C2) + 7.0f * (fabsf(A2) + fabsf(B2)));
    const float k3 = C3 + fmaxf(0.0f, 3.0f * A3) + fmaxf(0.0f, 3.0f * B3) - m + 1e-6f * (fabsf(C3) + 7.0f * (fabsf(A3) + fabsf(B3)));
    const float ax1 = 4.0f * A1, ax2 = 4.0f * A2, ax3 = 4.0f * A3, by1 = 4.0f * B1, by2 = 4.0f * B2, by3 = 4.0f * B3;
    s.ov[0] = x0 && y0 && k1 >= 0.0f && k2 >= 0.0f && k3 >= 0.0f;
    s.ov[1] = x1 && y0 && k1 + ax1 >= 0.0f && k2 + ax2 >= 0.0f && k3 + ax3 >= 0.0f;
    s.ov[2] = x0 && y1 && k1 + by1 >= 0.0f && k2 + by2 >= 0.0f && k3 + by3 >= 0.0f;
    s.ov[3] = x1 && y1 && k1 + ax1 + by1 >= 0.0f && k2 + ax2 + by2 >= 0.0f && k3 + ax3 + by3 >= 0.0f;
    return s;
}

// Row -1: a unit triangle a thousand pixels away with opacity 0 -> every pixel of the quadrant sees ecc ~ 3000 and alpha 0.
__device__ __forceinline__ void write_dummy_row(float *row, int lane)
{
    if (lane < ROW)
    {
        float v = 0.0f;
        if (lane == 0 || lane == 1 || lane == 3 || lane == 4) v = 1000.0f;
        if (lane == 2 || lane == 5) v = 1001.0f;
        if (lane == 6) v = 1.0f;
        if (lane == 18) v = __int_as_float(0x7fffffff); // list position of the dummy: beyond every pixel's range
        row[lane] = v;
    }
}

// The reference's per-pixel barycentrics (forward.cu:299-305, backward.cu:383-391): p_vk = v_k - pixel, a1 = cross(p_v2, p_v3) / area2,
// a2 = cross(p_v3, p_v1) / area2, a3 = 1 - a1 - a2, ecc = 1 - 3 min(a).  Division by area2 becomes a multiplication by its
// correctly rounded reciprocal (<= 1 ulp apart).
struct Bary { float p1x, p1y, p2x, p2y, p3x, p3y, a1, a2, a3, mn, ecc; };
__device__ __forceinline__ Bary barycentrics(const float4 &q0, const float4 &q1, float fx, float fy)
{
    Bary b;
    b.p1x = q0.x - fx; b.p1y = q0.y - fy; b.p2x = q0.z - fx; b.p2y = q0.w - fy; b.p3x = q1.x - fx; b.p3y = q1.y - fy;
    b.a1 = (b.p2x * b.p3y - b.p2y * b.p3x) * q1.z;
    b.a2 = (b.p3x * b.p1y - b.p3y * b.p1x) * q1.z;
    b.a3 = 1.0f - b.a1 - b.a2;
    b.mn = fminf(fminf(b.a1, b.a2), b.a3);
    b.ecc = fmaf(-3.0f, b.mn, 1.0f);
    return b;
}
#ifdef TS2D_STATS
// Profiling builds only (-DTS2D_STATS), read with ts2d_stats_read_group():
// [0] list entries visited (per quadrant wave)  [1] (entry, block) pairs surviving the cull  [2] wave steps  [3] windows
// [4] (pixel, entry) pairs blended  [5] quadrant waves  [6] batches with work  [7] (entry, quadrant) pairs surviving
__device__ unsigned long long g_stats_group[12];
#define TSG_STAT(i, v) stat_acc[i] += (unsigned long long)(v)
#else
#define TSG_STAT(i, v)
#endif

#if TSG_PART & 1
template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256, TSG_FWD_WAVES) render_fwd_group_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                                const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
                                                                float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                                float *__restrict__ out_feature, float *__restrict__ out_depth,
                                                                float *__restrict__ out_normal, float *__restrict__ contrib_sum,
                                                                float *__restrict__ contrib_max)
{
    __shared__ __attribute__((aligned(16))) float cst_all[4][(NR + 1) * ROW];
    __shared__ __attribute__((aligned(16))) uint32_t list_all[4][4 * NR / 2]; // per group: NR entries of (row | batch position << 8)
    // contrib_sum / contrib_max of the tile's first TCAP list entries, merged over the four quadrant waves before they leave
    // as global atomics (one L2 line operation per (tile, triangle) instead of one per (quadrant, triangle))
    constexpr int TCAP = TSG_TCAP; // 960: 960 x 12 bytes + the tables = 23.1 KB per workgroup: seven workgroups per CU (1024 entries would leave six)
    __shared__ unsigned long long tsum[RICH ? TCAP : 1]; // 16.48 fixed point
    __shared__ int tmax[RICH ? TCAP : 1];
#ifdef TSG_PAD_LDS // occupancy experiment: extra LDS bytes per workgroup
    __shared__ int pad_lds[TSG_PAD_LDS / 4];
    if (a.W < 0) pad_lds[threadIdx.x] = 1, atomicAdd(&tmax[0], pad_lds[255 - threadIdx.x]);
#endif

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, sub = lane & 15;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = ((grp & 1) << 2) + (sub & 3), ly = ((grp >> 1) << 2) + (sub >> 2);
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly, OX = (float)X0, OY = (float)Y0;
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    if (RICH)
    {
        for (int k = threadIdx.x; k < min(len, TCAP); k += 256) { tsum[k] = 0ull; tmax[k] = 0; }
        __syncthreads();
    }
    const float g2 = 2.0f * a.gamma;
    const float bg0 = a.background[0], bg1 = a.C > 1 ? a.background[1] : 0.0f, bg2 = a.C > 2 ? a.background[2] : 0.0f;
    float *cst = cst_all[wave] + ROW;
    uint32_t *list = list_all[wave];
    write_dummy_row(cst - ROW, lane);
    const char *lds0 = (const char *)cst_all;                                      // list entries are byte offsets from here
    const uint32_t row0 = (uint32_t)(wave * (NR + 1) + 1) * (ROW * 4), dummy = row0 - ROW * 4; // row r of this wave: row0 + r * 80
    const int stat_step = ((lane >> 3) & 1) | ((lane >> 1) & 2) | ((lane << 1) & 4); // the step of a window whose statistics this lane ends up with

    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, anx = 0.0f, any_ = 0.0f, anz = 0.0f, ad = 0.0f;
    bool done = !inside;
    uint32_t last = (uint32_t)len; // a pixel that never saturates examines the whole list (forward.cu:296-297)

#ifdef TS2D_STATS
    unsigned long long stat_acc[12] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
#endif
    // dense batches: only the entries whose quadrant bit is set are gathered and culled (ts2d_group.h, stream_refill); `pos` = list position
    uint32_t id = 0;
    int pos = 0, cursor = 0, carried = 0; // carried: entries of the previous batch that did not fit its table (lanes [0, carried) of id / pos)
    for (;;)
    {
        const unsigned long long alive = ballot(!done);
        if (alive == 0) break;
        int nq = TSG_CARRY ? carried : 0;
        carried = 0;
        stream_refill<false, TSG_FWD_CAP>(id, pos, nq, point_list + range.x, cursor, len, TS_ID_BITS + wave, lane);
        if (nq == 0) break;
        const bool valid = lane < nq;
        const int ent = pos;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (valid)
        {
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            if (RICH) r3 = rp[3];
        }
        const BlockCull s = block_cull<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY);
        // one entry mask per block; a block whose 16 pixels are all saturated takes no more entries
        unsigned long long M[4];
#pragma unroll
        for (int g = 0; g < 4; g++) M[g] = ((alive >> (16 * g)) & 0xFFFFull) ? ballot(valid && s.ov[g]) : 0ull;
        const unsigned long long any = M[0] | M[1] | M[2] | M[3];
        TSG_STAT(0, __popcll(ballot(valid)));
        if (any == 0) continue;
        TSG_STAT(1, __popcll(M[0]) + __popcll(M[1]) + __popcll(M[2]) + __popcll(M[3]));
        TSG_STAT(6, 1);
        TSG_STAT(7, __popcll(any));
        // The entries with work are COMPACTED into at most NR table rows per pass (a batch with more survivors takes two passes):
        // half the LDS of a row per list entry, hence 7 instead of 5 resident waves per SIMD -- the blend kernels are latency
        // bound (3 instead of 5 waves: +27 %, profiles/r02_notes.md).
        const bool anybit = (any >> lane) & 1;
        const int rank = lane_rank(any), nact = __popcll(any);
        const int r = rank & (NR - 1);
        bool mine = anybit && rank < NR;
        if (mine) publish_row(cst + r * ROW, s, id, ent, r1, r2, r3);
        for ([[maybe_unused]] int h = 0;;)
        {
            const unsigned long long mm = nact <= NR ? any : ballot(mine);
            list[lane] = dummy | (dummy << 16); // four lists x NR entries: the dummy row
            int steps = 0;
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                const unsigned long long Mh = M[g] & mm;
                if ((Mh >> lane) & 1) ((u16a *)list)[g * NR + lane_rank(Mh)] = (unsigned short)(row0 + r * (ROW * 4));
                steps = max(steps, __popcll(Mh));
            }
            const u16a *mylist = (const u16a *)list + grp * NR;
            TSG_STAT(2, steps);
            TSG_STAT(3, (steps + 7) / 8);
#ifdef TS2D_STATS
            {   // [8] steps at which two groups hold the same entry (what the backward must serialise)  [9] passes  [10] second passes
                const u16a *l16 = (const u16a *)list + (lane & (NR - 1));
                const uint32_t l0 = l16[0], l1 = l16[NR], l2 = l16[2 * NR], l3 = l16[3 * NR];
                TSG_STAT(8, __popcll(ballot(lane < NR && ((l0 != dummy && (l0 == l1 || l0 == l2 || l0 == l3)) || (l1 != dummy && (l1 == l2 || l1 == l3)) ||
                                                          (l2 != dummy && l2 == l3)))));
                TSG_STAT(9, 1);
                TSG_STAT(10, h > 0 ? 1 : 0);
            }
#endif

            for (int t0 = 0; t0 < steps; t0 += 8)
            {
                float c[8];
                const uint4 packed = *(const uint4 *)(mylist + t0); // this group's next 8 entries (row byte offsets)
#pragma unroll
                for (int st = 0; st < 8; st++)
                {
                    c[st] = 0.0f;
                    if (t0 + st < steps)
                    {
                        const uint32_t word = st < 2 ? packed.x : (st < 4 ? packed.y : (st < 6 ? packed.z : packed.w));
                        const float *row = (const float *)(lds0 + ((st & 1) ? (word >> 16) : (word & 0xFFFFu)));
                        const float4 q0 = *(const float4 *)(row), q1 = *(const float4 *)(row + 4);
                        const Bary b = barycentrics(q0, q1, fx, fy);
                        const float4 q2 = *(const float4 *)(row + 8);
                        float4 q3 = make_float4(0, 0, 0, 0);
                        float vd3 = 0.0f;
                        int jpos; // position in the tile's list
                        if (RICH)
                        {
                            q3 = *(const float4 *)(row + 12);
                            const float4 q4 = *(const float4 *)(row + 16);
                            vd3 = q4.x;
                            jpos = __float_as_int(q4.z);
                        }
                        else jpos = __float_as_int(row[18]);
                        const float pw = GAMMA1 ? b.ecc * b.ecc : pow_nonneg(b.ecc, g2);
                        const float alpha = fminf(0.99f, q1.w * __builtin_amdgcn_exp2f(pw * -0.7213475204444817f)); // forward.cu:311-312
                        const bool hit = !done && ecc_in_range(b.ecc) && alpha >= 1.0f / 255.0f;                     // forward.cu:307,313
                        // branch-free blend: a lane that does not hit runs with alpha = 0 (x + c*0 == x, T*1 == T bit for bit)
                        const float al = hit ? alpha : 0.0f;
                        TSG_STAT(4, __popcll(ballot(hit)));
                        const float contrib = al * T;
                        ar = fmaf(q2.x, contrib, ar);
                        ag = fmaf(q2.y, contrib, ag);
                        ab = fmaf(q2.z, contrib, ab);
                        if (RICH)
                        {
                            anx = fmaf(q2.w, contrib, anx);
                            any_ = fmaf(q3.x, contrib, any_);
                            anz = fmaf(q3.y, contrib, anz);
                            const float d = q3.z * b.a1 + q3.w * b.a2 + vd3 * b.a3; // forward.cu:328
                            ad = fmaf(d, contrib, ad);
                            c[st] = contrib;
                        }
                        T *= (1.0f - al);
                        const bool sat = hit && T <= 0.0001f; // forward.cu:333
                        last = sat ? (uint32_t)(jpos + 1) : last;
                        done = done || sat;
                    }
                }
#if TSG_PROBE != 2
                if (RICH)
                {
                    // contrib_sum / contrib_max (forward.cu:323-324; the reference issues two global atomics per (pixel, triangle)):
                    // the window's 8 x 64 contributions are reduced inside each 16-lane group, lane pairs (l, l ^ 1) end up with
                    // (sum, max, batch position) of step `b3 + 2 b2 + 4 b1` of their group, and the even lanes add them to the TILE's
                    // statistics in LDS with INTEGER atomics (ts2d_group.h: ds_add_u64 / ds_max_i32 cost 5-7 cycles per wave
                    // instruction, ds_add_f32 193) -- no ordering between groups or waves is needed.
                    float sm, mx;
                    row_reduce8_sum_max(c, 0xCCCCCCCCCCCCCCCCull, sm, mx);
                    // the list position of "its" step: from the step's row (two LDS reads on the few lanes that have something to add) rather
                    // than carried through the window in eight registers and selected with seven v_cndmask
                    int k = 0;
                    if ((lane & 1) == 0 && sm > 0.0f) k = __float_as_int(*(const float *)(lds0 + mylist[t0 + stat_step] + 18 * 4));
                    if ((lane & 1) == 0 && sm > 0.0f) tile_stats_add<TCAP>(tsum, tmax, k, sm, mx, point_list + range.x, contrib_sum, contrib_max);
                }
#endif
            }
#if TSG_CARRY
            // more survivors than table rows: the ones beyond the table open the NEXT batch (one more gather + cull of a few lanes that the batch
            // runs anyway) instead of a second pass of their own with its list build and its short lockstep loop -- group_sim: -26 % passes
            if (nact > NR) carried = carry_over(id, pos, anybit && rank >= NR, lane);
            break;
#else
            if (++h * NR >= nact) break;
            mine = anybit && rank >= NR;
            if (mine) republish_row<RICH>(cst + r * ROW, point_list, rec, range.x + pos, ent, OX, OY);
#endif
        }
    }

#ifdef TS2D_STATS
    if (lane == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&g_stats_group[i], stat_acc[i]);
#endif
    // the wave's pixels leave first: their stores, and the ids the flush below needs, are in flight while the wave waits for the others
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
    if (RICH)
    {
        constexpr int NF = (TCAP + 255) / 256;
        const int nflush = min(len, TCAP);
        uint32_t ids[NF];
#pragma unroll
        for (int j = 0; j < NF; j++)
        {
            const int k = (int)threadIdx.x + 256 * j;
            ids[j] = k < nflush ? point_list[range.x + k] & TS_ID_MASK : 0u;
        }
        __syncthreads(); // the only rendezvous of the four quadrant waves: the tile's merged contribution statistics leave
#pragma unroll
        for (int j = 0; j < NF; j++)
        {
            const int k = (int)threadIdx.x + 256 * j;
            if (k < nflush)
            {
                const unsigned long long fx48 = tsum[k];
                if (TSG_PROBE != 5 && fx48 != 0ull) tile_stats_flush(fx48, tmax[k], TSG_PROBE == 9 ? ((range.x + (uint32_t)k) & 0x3FFFFu) : ids[j], contrib_sum, contrib_max);
            }
        }
    }
}

#endif // TSG_PART & 1

// Backward.  Per (pixel, triangle) pair the reference adds 16 values into per-triangle arrays (backward.cu:412-490); here the
// pair's 16 values are formed per lane exactly in the reference's per-pixel form (no moment / epilogue algebra):
//   dL/dv_j (screen space) = perp(t_j) / area2 with  t_1 = e_3 p_v2 - e_2 p_v3,  t_2 = e_1 p_v3 - e_3 p_v1,  t_3 = e_2 p_v1 - e_1 p_v2,
//   e_k = dL/da_k - sum_m dL/da_m a_m     (backward.cu:464-479 regrouped: v2_v3 = p_v3 - p_v2 etc.; perp(x, y) = (y, -x)),
// the division by area2 is applied once per entry when the sums are flushed.  The reference's seven back-to-front
// composites per pixel collapse to one scalar B = sum_c dL_dpix_c * accum_c (same mathematics, see render.hip).
// Each group reduces its 16 values over its 16 lanes (DPP row transpose-reduce) and adds them into the entry's row of a
// wave-private LDS table (one group after the other: two groups may be working on the same entry); once per batch the rows
// leave as coalesced 64-byte atomic adds, one gradient record per 16 lanes.
// WPB = quadrant waves per workgroup.  The four quadrant waves of a tile never talk to each other here, so the backward launches
// them as single-wave workgroups (WPB = 1): the dispatcher then fills a freed wave slot with the next quadrant instead of waiting for
// four slots of one CU, which shortens the tail of the launch (8160 tiles are only 5.3 rounds of 256-thread workgroups).  The four
// quadrants of a tile stay neighbours in dispatch order and on one XCD (shared L2 for the tile's list and records).
#if TSG_PART & 2
template <bool RICH, bool GAMMA1, int WPB>
__global__ void __launch_bounds__(64 * WPB, TSG_BWD_WAVES) render_bwd_group_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                                   const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
                                                                   const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                                   const float *__restrict__ dL_dout_feature,
                                                                   const float *__restrict__ dL_dout_depth,
                                                                   const float *__restrict__ dL_dout_normal, float *__restrict__ grad_rec)
{
    __shared__ __attribute__((aligned(16))) float rows_all[WPB][(NR + 1) * BROW]; // constants + gradient sums; row -1 absorbs the adds of idle groups
    __shared__ __attribute__((aligned(16))) uint32_t list_all[WPB][4 * NR / 2];   // per group: NR entries (u16 byte offsets of rows)
#ifdef TSG_PAD_LDS
    __shared__ int pad_lds_b[TSG_PAD_LDS / 4];
    if (a.W < 0) pad_lds_b[threadIdx.x] = 1, list_all[0][0] = (signed char)pad_lds_b[255 - threadIdx.x];
#endif

    int tile, quad, wave; // quad = which 8x8 quadrant of the tile, wave = index into this workgroup's LDS arrays
    if (WPB == 4)
    {
        tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
        quad = wave = threadIdx.x >> 6;
    }
    else
    {
        // single-wave workgroups: four consecutive units of an XCD are the four quadrants of one tile (they stay neighbours in dispatch order
        // and on one XCD: shared L2 for the tile's list and records)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = tile_of_block(((j >> 2) << 3) | x, a.grid_x, a.grid_y);
        quad = j & 3;
        wave = 0;
    }
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4, sub = lane & 15;
    const int X0 = tx * TS_TILE + (quad & 1) * 8, Y0 = ty * TS_TILE + (quad >> 1) * 8;
    const int lx = ((grp & 1) << 2) + (sub & 3), ly = ((grp >> 1) << 2) + (sub >> 2);
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly, OX = (float)X0, OY = (float)Y0;
    const uint2 range = ranges[tile];
    const float g2 = 2.0f * a.gamma;
    const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
    float *rows = rows_all[wave] + BROW;
    uint32_t *list = list_all[wave];
    write_dummy_row(rows - BROW, lane);
    char *lds0 = (char *)rows_all;                                                   // list entries are byte offsets from here
    const uint32_t row0 = (uint32_t)(wave * (NR + 1) + 1) * (BROW * 4), dummy = row0 - BROW * 4; // row r of this wave: row0 + r * 144
    const uint32_t accoff = ROW * 4 + 4 * sub;                                        // this lane's sum inside a row

    float T = inside ? final_T[pix] : 0.0f;            // backward.cu:318
    const int last = inside ? (int)n_contrib[pix] : 0; // backward.cu:320
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
    // The six colour / normal columns of the per-step reduction are (dL_dpixel constant) x contrib: their registers are filled
    // pre-swapped (ts2d_group.h, row_reduce16c) with constants that depend on which quarter of its 16-lane group the lane is in.
    // Quad (registers 0..3) = r, b, g, nx by final position; pair (registers 4, 5) = ny, nz.
    const int quarter = sub >> 2;
    const float kq0 = quarter == 0 ? dpr : (quarter == 1 ? dpg : (quarter == 2 ? dpb : dnx)); // [X0 X1 Y0 Y1], X0 = r, X1 = g, Y0 = b, Y1 = nx
    const float kq1 = quarter == 0 ? dpb : (quarter == 1 ? dnx : (quarter == 2 ? dpr : dpg)); // [Y0 Y1 X0 X1]
    const float kq2 = quarter == 0 ? dpg : (quarter == 1 ? dpr : (quarter == 2 ? dnx : dpb)); // [X1 X0 Y1 Y0]
    const float kq3 = quarter == 0 ? dnx : (quarter == 1 ? dpb : (quarter == 2 ? dpg : dpr)); // [Y1 Y0 X1 X0]
    const float kp4 = (sub & 8) ? dnz : dny, kp5 = (sub & 8) ? dny : dnz;
    // gradient-record column (0..5 screen vertices, 6 opacity, 7..9 rgb, 10..12 normal, 13..15 vertex depths) of the value lane `sub`
    // ends up with: register reg(sub) of the network, registers -> columns {7, 9, 8, 10, 11, 12, 0, 1, 2, 3, 4, 5, 6, 13, 14, 15}
    const int rcol = (int)((0xF15ADC39E0486B27ull >> (4 * sub)) & 15ull);
    // entries at list positions >= the largest n_contrib of a block are skipped by all of its pixels (backward.cu:377-379)
    float lm = (float)last;
    lm = fmaxf(lm, dpp<DPP_XOR1>(lm));
    lm = fmaxf(lm, dpp<DPP_XOR2>(lm));
    lm = fmaxf(lm, dpp<DPP_HALF_MIRROR>(lm));
    lm = fmaxf(lm, dpp<DPP_MIRROR>(lm));
    int glast[4];
#pragma unroll
    for (int g = 0; g < 4; g++) glast[g] = (int)__builtin_amdgcn_readlane((int)lm, 16 * g);
    const int maxlast = max(max(glast[0], glast[1]), max(glast[2], glast[3]));
    if (maxlast <= 0) return;

    // dense batches, walked back to front: lane 0 holds the entry farthest back (ts2d_group.h, stream_refill<true>); `pos` = list position
    uint32_t id = 0;
    int pos = 0, cursor = maxlast, carried = 0;
    for (;;)
    {
        int nq = TSG_CARRY ? carried : 0;
        carried = 0;
        stream_refill<true, TSG_BWD_CAP>(id, pos, nq, point_list + range.x, cursor, maxlast, TS_ID_BITS + quad, lane);
        if (nq == 0) break;
        const bool valid = lane < nq;
        const int ent = pos;
        float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
        if (valid)
        {
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
            if (RICH) r3 = rp[3];
        }
        const BlockCull s = block_cull<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY);
        unsigned long long M[4];
#pragma unroll
        for (int g = 0; g < 4; g++) M[g] = ballot(valid && s.ov[g] && pos < glast[g]); // entries at or behind glast[g] are skipped by all of block g's pixels
        const unsigned long long any = M[0] | M[1] | M[2] | M[3];
        if (any == 0) continue;
        // compacted table rows, at most NR per pass (see the forward); back to front = the low lanes first (lane 0 is the entry farthest back)
        const bool anybit = (any >> lane) & 1;
        const int rank = lane_rank(any), nact = __popcll(any);
        const int r = rank & (NR - 1);
        bool mine = anybit && rank < NR;
        if (mine) publish_row(rows + r * BROW, s, id, ent, r1, r2, r3);
        for ([[maybe_unused]] int h = (nact - 1) / NR;;)
        {
            const unsigned long long mm = nact <= NR ? any : ballot(mine);
            if (mine)
            {
                float4 *z = (float4 *)(rows + r * BROW + ROW);
                z[0] = z[1] = z[2] = z[3] = make_float4(0, 0, 0, 0);
            }
            list[lane] = dummy | (dummy << 16);
            int steps = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) // back to front: the entry with the highest list position first
            {
                const unsigned long long Mh = M[g] & mm;
                const int n = __popcll(Mh);
                if ((Mh >> lane) & 1) ((u16a *)list)[g * NR + lane_rank(Mh)] = (unsigned short)(row0 + r * (BROW * 4));
                steps = max(steps, n);
            }
            const u16a *mylist = (const u16a *)list + grp * NR;
            if (TSG_PROBE == 4) steps = 0;

            // steps at which two groups work on the SAME entry (their sums must then be added to its row one after the other)
            unsigned long long conflict;
            {
                const u16a *l16 = (const u16a *)list + (lane & (NR - 1));
                const uint32_t l0 = l16[0], l1 = l16[NR], l2 = l16[2 * NR], l3 = l16[3 * NR];
                conflict = ballot(lane < NR && ((l0 != dummy && (l0 == l1 || l0 == l2 || l0 == l3)) || (l1 != dummy && (l1 == l2 || l1 == l3)) ||
                                                (l2 != dummy && l2 == l3)));
            }
            uint32_t ra_next = mylist[0];
            for (int t0 = 0; t0 < steps; t0++)
            {
                {
                    const uint32_t ra = ra_next;    // this group's next entry; past the end of its list: the dummy row
                    ra_next = mylist[min(t0 + 1, NR - 1)]; // fetched one step ahead: one LDS round trip less on the step's critical path
                    const float *row = (const float *)(lds0 + ra);
                    float *acc = (float *)(lds0 + ra + accoff);
                    const bool shared_row = TSG_PROBE == 3 ? false : (bool)((conflict >> t0) & 1); // wave-uniform
                    const float q0acc = *acc;                              // fetched early; only used when no other group adds to this row now
                    const float4 q0 = *(const float4 *)(row), q1 = *(const float4 *)(row + 4);
                    const Bary b = barycentrics(q0, q1, fx, fy);
                    const float4 q2 = *(const float4 *)(row + 8);
                    float4 q3 = make_float4(0, 0, 0, 0);
                    float vd3 = 0.0f;
                    int jpos; // position in the tile's list
                    if (RICH)
                    {
                        q3 = *(const float4 *)(row + 12);
                        const float4 q4 = *(const float4 *)(row + 16);
                        vd3 = q4.x;
                        jpos = __float_as_int(q4.z);
                    }
                    else jpos = __float_as_int(row[18]);
                    const float pw = GAMMA1 ? b.ecc * b.ecc : pow_nonneg(b.ecc, g2);
                    const float G = __builtin_amdgcn_exp2f(pw * -0.7213475204444817f); // exp(-0.5 pw)
                    const float opG = q1.w * G;
                    const float alpha = fminf(0.99f, opG);
                    const bool hit = (jpos < last) && ecc_in_range(b.ecc) && alpha >= 1.0f / 255.0f; // backward.cu:378,393,400
                    // branch-free from here on: a lane that does not hit runs with alpha = 0, so T and B stay bit-unchanged and every
                    // value it feeds into the reduction is an exact 0
                    const float al = hit ? alpha : 0.0f;
                    const float oma = 1.0f - al;
                    T = T * __builtin_amdgcn_rcpf(oma); // backward.cu:403
                    const float contrib = al * T;
                    float X = fmaf(dpb, q2.z, fmaf(dpg, q2.y, dpr * q2.x)); // backward.cu:415
                    float w = 0.0f;
                    if (RICH) // backward.cu:419-437
                    {
                        X = fmaf(dnz, q3.y, fmaf(dny, q3.x, fmaf(dnx, q2.w, X)));
                        const float depth = fmaf(vd3, b.a3, fmaf(q3.w, b.a2, q3.z * b.a1));
                        X = fmaf(dd, depth, X);
                        w = dd * contrib; // dL_ddepth
                    }
                    const float dL_dcontrib = X - B;
                    B = fmaf(al, X, oma * B);
                    const float dL_dalpha = dL_dcontrib * T;
                    // backward.cu:443-447: dL_decc = dL_dpower * 2 gamma * power / (ecc + 1e-8) with power = -0.5 pw and
                    // dL_dpower = dL_dalpha * alpha unless the 0.99 clamp was active; z = -3 dL_decc goes to the arg-min barycentric
                    // (for gamma = 1, pw / (ecc + 1e-8) is ecc to 1e-8 / ecc relative, and a pair with ecc that small contributes ~ecc to begin
                    // with: one multiplication instead of a reciprocal)
                    const float zr = GAMMA1 ? 1.5f * g2 * (dL_dalpha * alpha) * b.ecc
                                            : 1.5f * g2 * (dL_dalpha * alpha) * pw * __builtin_amdgcn_rcpf(b.ecc + 1e-8f);
                    const float z = (hit && opG < 0.99f) ? zr : 0.0f; // the select sits last: a lane that does not hit may hold inf / NaN in pw
                    const bool k1 = b.a1 == b.mn;        // backward.cu:449-461: a1 <= a2 && a1 <= a3, then a2 <= a1 && a2 <= a3, else a3
                    const bool k2 = !k1 && b.a2 == b.mn;
                    const float z1 = k1 ? z : 0.0f, z2 = k2 ? z : 0.0f, z3 = z - z1 - z2;
                    const float da1 = fmaf(w, q3.z, z1), da2 = fmaf(w, q3.w, z2), da3 = fmaf(w, vd3, z3); // backward.cu:433,462
                    const float sdot = fmaf(da3, b.a3, fmaf(da2, b.a2, da1 * b.a1));
                    const float e1 = da1 - sdot, e2 = da2 - sdot, e3 = da3 - sdot;
                    float v[16];
                    v[0] = kq0 * contrib; v[1] = kq1 * contrib; v[2] = kq2 * contrib; v[3] = kq3 * contrib; // dL/drgb, dL/dn.x (backward.cu:412, 421)
                    v[4] = kp4 * contrib; v[5] = kp5 * contrib;                                             // dL/dn.y, dL/dn.z (:422-423)
                    v[6] = e3 * b.p2y - e2 * b.p3y;  // perp(t_1).x =  t_1.y
                    v[7] = e2 * b.p3x - e3 * b.p2x;  // perp(t_1).y = -t_1.x
                    v[8] = e1 * b.p3y - e3 * b.p1y;
                    v[9] = e3 * b.p1x - e1 * b.p3x;
                    v[10] = e2 * b.p1y - e1 * b.p2y;
                    v[11] = e1 * b.p2x - e2 * b.p1x;
                    v[12] = hit ? dL_dalpha * G : 0.0f; // backward.cu:490 (not gated by the clamp)
                    v[13] = w * b.a1; v[14] = w * b.a2; v[15] = w * b.a3; // backward.cu:429-431
                    const float red = row_reduce16c(v, 0xCCCCCCCCCCCCCCCCull, 0xAAAAAAAAAAAAAAAAull); // lane sub: gradient-record column rcol
                    if (!shared_row) *acc = q0acc + red;
                    else
                    {
#pragma unroll
                        for (int g = 0; g < 4; g++)
                        {
                            if (grp == g) *acc += red;
                            wave_lds_order();
                        }
                    }
                }
            }

            // Pass flush: 16 consecutive lanes add the 16 floats (one 64-byte line) of one triangle's gradient record, four
            // rows per instruction; the vertex columns get their 1 / area2 here.
            {
                const int n = __popcll(mm);
#pragma unroll 1
                for (int e0 = 0; e0 < n; e0 += 4)
                {
                    const int e = e0 + grp;
                    if (e < n)
                    {
                        const uint32_t eid = __float_as_uint(rows[e * BROW + 17]);
                        float val = rows[e * BROW + ROW + sub];
                        if (rcol < 6) val *= rows[e * BROW + 6];
#if TSG_PROBE == 6
                        if (a.W < 0) grad_rec[eid] = val; // never taken: keeps `val` alive
#elif TSG_PROBE == 7
                        if (RICH || rcol < 10) grad_rec[TS_GRAD_FLOATS * (size_t)eid + rcol] = val;
#else
                        if (RICH || rcol < 10) unsafeAtomicAdd(grad_rec + TS_GRAD_FLOATS * (size_t)eid + rcol, val);
#endif
                    }
                }
            }
#if TSG_CARRY
            if (nact > NR) carried = carry_over(id, pos, anybit && rank >= NR, lane); // see the forward
            break;
#else
            if (--h < 0) break;
            mine = anybit && rank >= NR;
            if (mine) republish_row<RICH>(rows + r * BROW, point_list, rec, range.x + pos, ent, OX, OY);
#endif
        }
    }
}
#endif // TSG_PART & 2
} // namespace

#if TSG_PART & 1
#define TS_DISPATCH_G(KERNEL, ...)                                                                                    \
    do                                                                                                                \
    {                                                                                                                 \
        const bool g1 = (a.gamma == 1.0f);                                                                            \
        if (a.rich_info && g1) hipLaunchKernelGGL((KERNEL<true, true>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (a.rich_info) hipLaunchKernelGGL((KERNEL<true, false>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (g1) hipLaunchKernelGGL((KERNEL<false, true>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, dim3(256), 0, s, __VA_ARGS__);                          \
    } while (0)

void ts_launch_render_fwd_group(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                                float *out_feature, float *out_depth, float *out_normal, float *contrib_sum, float *contrib_max,
                                hipStream_t s)
{
    if (a.grid_x * a.grid_y == 0) return;
    const dim3 grid((unsigned)ts_tile_units(a.grid_x, a.grid_y));
    TS_DISPATCH_G(render_fwd_group_kernel, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature, out_depth, out_normal,
                  contrib_sum, contrib_max);
}

#endif // TSG_PART & 1

#if defined(TS2D_STATS) && (TSG_PART & 1)
extern "C" __attribute__((visibility("default"))) int ts2d_stats_read_group(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats_group), sizeof(unsigned long long) * 12);
    if (e == hipSuccess && reset)
    {
        unsigned long long z[12] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stats_group), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : 2;
}
#endif

#if TSG_PART & 2
void ts_launch_render_bwd_group(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                                const float *dL_dout_feature, const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec,
                                hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    constexpr int WPB = 1;
    const dim3 grid1((unsigned)((WPB == 4 ? 1 : 4) * ts_tile_units(a.grid_x, a.grid_y))); // padded: units past the image return at once
    const bool g1 = (a.gamma == 1.0f);
#define TS_BWD(R, G) hipLaunchKernelGGL((render_bwd_group_kernel<R, G, WPB>), grid1, dim3(64 * WPB), 0, s, a, im.ranges, b.vals, g.rec, im.final_T, \
                                        im.n_contrib, dL_dout_feature, dL_dout_depth, dL_dout_normal, grad_rec)
    if (a.rich_info && g1) TS_BWD(true, true);
    else if (a.rich_info) TS_BWD(true, false);
    else if (g1) TS_BWD(false, true);
    else TS_BWD(false, false);
#undef TS_BWD
}
#endif // TSG_PART & 2
