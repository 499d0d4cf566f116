// ts2d_group.h -- building blocks shared by the lane-group blend kernels (render_group.hip: 2D, render3d_group.hip: 3D):
// wave64 ballots / ranks, the 16-lane DPP-row transpose-reduce networks, the ecc range test.
#pragma once
#include "ts2d_wave.h"

namespace
{
constexpr int NR = 32;  // table rows per pass: the entries of a 64-entry batch that have work, compacted (more than NR: two passes)
typedef unsigned short __attribute__((may_alias)) u16a;
constexpr int ROW = 20; // floats per entry row of the wave-private constants table (row -1 = a dummy no pixel can hit)

__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int lane_rank(unsigned long long m) // set bits of m below this lane
{
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// 0 <= ecc <= 10 (forward.cu:307) as ONE unsigned compare: negative floats and NaN have larger bit patterns than 10.0f
__device__ __forceinline__ bool ecc_in_range(float ecc) { return __float_as_uint(ecc) <= 0x41200000u; }

// ---- dense batches -------------------------------------------------------------------------------------------------------
// The emission kernel marks in the top four bits of an instance's value which quadrants of the tile the triangle can reach (ts2d_support.h).
// A quadrant wave reads the tile's list 64 candidates at a time and keeps the entries whose bit is set, COMPACTED: lanes [0, n) of (id, pos)
// hold the next n entries in processing order (pos = the entry's position in the tile's list).  One refill = one ballot, one cross-lane
// permutation (ds_permute: taken lanes go to slots n, n + 1, ..., the others fill the rest of the permutation) and no LDS memory; a window
// that does not fit is consumed only up to the last entry that does (the cursor moves by that many candidates).
// DESC = false: candidates cursor, cursor + 1, ... (< end), cursor moves up.  DESC = true: cursor - 1, cursor - 2, ... (>= 0), cursor moves down
// (the backward walks the list back to front; lane 0 is then the entry farthest back).
// CAP = entries per batch (64; 32 = every batch is ONE pass of the 32-row table and no record is gathered twice -- measured in round 5,
// profiles/r05_notes.md: the extra block culls cost what the re-gathers save).
template <bool DESC, int CAP = 64>
__device__ __forceinline__ void stream_refill(uint32_t &id, int &pos, int &n, const uint32_t *__restrict__ list, int &cursor, int end, int qbit, int lane)
{
    while (n < CAP && (DESC ? cursor > 0 : cursor < end))
    {
        const int k = DESC ? cursor - 1 - lane : cursor + lane;
        const bool valid = DESC ? k >= 0 : k < end;
        const uint32_t v = valid ? list[k] : 0u;
        const bool want = valid && ((v >> qbit) & 1u);
        unsigned long long mt = ballot(want);
        int adv = 64;
        const int room = CAP - n;
        bool taken = want;
        if (__popcll(mt) > room) // keep the first `room` of them; the next refill starts behind the last one kept
        {
            taken = want && lane_rank(mt) < room;
            mt = ballot(taken);
            adv = 64 - __builtin_clzll(mt);
        }
        cursor += DESC ? -adv : adv;
        if (mt == 0) continue;
        const int cnt = __popcll(mt), rk = lane_rank(mt);
        const int dest = (taken ? n + rk : n + cnt + (lane - rk)) & 63; // a permutation of the 64 lanes
        const uint32_t pid = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)(v & TS_ID_MASK));
        const int ppos = __builtin_amdgcn_ds_permute(dest << 2, k);
        const bool fresh = lane >= n && lane < n + cnt;
        id = fresh ? pid : id;
        pos = fresh ? ppos : pos;
        n += cnt;
    }
}

// The entries of a batch flagged `keep` move to lanes [0, n) of (id, pos), in lane order (= processing order); returns n.  One ballot, one
// permutation of the 64 lanes -- the same move as stream_refill's.  The next stream_refill appends behind them.
__device__ __forceinline__ int carry_over(uint32_t &id, int &pos, bool keep, int lane)
{
    const unsigned long long mk = ballot(keep);
    const int cnt = __popcll(mk), rk = lane_rank(mk);
    const int dest = keep ? rk : cnt + (lane - rk); // kept lanes to the front, the others behind them: a permutation
    id = (uint32_t)__builtin_amdgcn_ds_permute(dest << 2, (int)id);
    pos = __builtin_amdgcn_ds_permute(dest << 2, pos);
    return cnt;
}

// ---- 16-lane (DPP row) transpose-reduce: N values per lane -> each lane keeps the row-wide reduction of ONE value ----
// Level 1 pairs lanes l, l ^ 8 (row_ror:8), level 2 lanes inside a group of 8 (row_half_mirror), level 3 l, l ^ 2, then l, l ^ 1.
struct RowSel
{
    bool b3, b2, b1;
    __device__ __forceinline__ explicit RowSel(int lane) : b3(lane & 8), b2(lane & 4), b1(lane & 2) {}
};
template <typename Op>
__device__ __forceinline__ float pair_ror8(float x, float y, bool b, Op op)
{
    const float own = b ? y : x, oth = b ? x : y;
    return op(own, dpp<DPP_ROR8>(oth));
}
template <typename Op>
__device__ __forceinline__ float pair_hmir(float x, float y, bool b, Op op)
{
    const float own = b ? y : x, oth = b ? x : y;
    return op(own, dpp<DPP_HALF_MIRROR>(oth));
}
template <typename Op>
__device__ __forceinline__ float pair_xor2(float x, float y, bool b, Op op)
{
    const float own = b ? y : x, oth = b ? x : y;
    return op(own, dpp<DPP_XOR2>(oth));
}
// 8 values: the result of value (b3 + 2 b2 + 4 b1) lands in lanes l and l ^ 1
template <typename Op>
__device__ __forceinline__ float row_reduce8(const float (&c)[8], const RowSel &r, Op op)
{
    const float s0 = pair_ror8(c[0], c[1], r.b3, op), s1 = pair_ror8(c[2], c[3], r.b3, op);
    const float s2 = pair_ror8(c[4], c[5], r.b3, op), s3 = pair_ror8(c[6], c[7], r.b3, op);
    const float t0 = pair_hmir(s0, s1, r.b2, op), t1 = pair_hmir(s2, s3, r.b2, op);
    const float v = pair_xor2(t0, t1, r.b1, op);
    return op(v, dpp<DPP_XOR1>(v));
}
// Sum AND maximum of 8 values over the 16 lanes of each DPP row in one pass (the forward's contribution statistics): the transposed network of
// row_reduce8 with the write masks of row_reduce16 -- levels 1 and 2 pair lanes of different DPP banks, so "which half keeps which value" is the
// instruction's own bank mask (two v_add/v_max_f32_dpp per pair instead of two v_cndmask + one; every one of them half rate on gfx950): 32
// instructions for both results against 44.  The two chains are interleaved so that no DPP operand is read within two wait states of its
// write; level 1 writes fresh registers (the inputs stay intact for the caller).  Lanes l and l ^ 1 end up with step (b3 + 2 b2 + 4 b1).
__device__ __forceinline__ void row_reduce8_sum_max(const float (&c)[8], unsigned long long mask_b1, float &sum, float &mx)
{
    float s0, s1, s2, s3, m0, m1, m2, m3;
#define TSG8_L1(OP, T, X, Y)                                                              \
    OP " " T ", " Y ", " Y " row_ror:8 row_mask:0xf bank_mask:0xc\n"                      \
    OP " " T ", " X ", " X " row_ror:8 row_mask:0xf bank_mask:0x3\n"
#define TSG8_L2(OP, X, Y)                                                                 \
    OP " " Y ", " Y ", " Y " row_half_mirror row_mask:0xf bank_mask:0xa\n"                \
    OP " " Y ", " X ", " X " row_half_mirror row_mask:0xf bank_mask:0x5\n"
#define TSG8_QP(OP, X, QP) OP " " X ", " X ", " X " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"
    asm volatile("s_nop 1\n"
                 TSG8_L1("v_add_f32_dpp", "%0", "%8", "%9") TSG8_L1("v_add_f32_dpp", "%1", "%10", "%11")
                 TSG8_L1("v_add_f32_dpp", "%2", "%12", "%13") TSG8_L1("v_add_f32_dpp", "%3", "%14", "%15")
                 TSG8_L1("v_max_f32_dpp", "%4", "%8", "%9") TSG8_L1("v_max_f32_dpp", "%5", "%10", "%11")
                 TSG8_L1("v_max_f32_dpp", "%6", "%12", "%13") TSG8_L1("v_max_f32_dpp", "%7", "%14", "%15")
                 TSG8_L2("v_add_f32_dpp", "%0", "%1") TSG8_L2("v_max_f32_dpp", "%4", "%5")
                 TSG8_L2("v_add_f32_dpp", "%2", "%3") TSG8_L2("v_max_f32_dpp", "%6", "%7")
                 TSG8_QP("v_add_f32_dpp", "%1", "[2,3,0,1]") TSG8_QP("v_max_f32_dpp", "%5", "[2,3,0,1]")
                 TSG8_QP("v_add_f32_dpp", "%3", "[2,3,0,1]") TSG8_QP("v_max_f32_dpp", "%7", "[2,3,0,1]")
                 "v_cndmask_b32_e64 %3, %1, %3, %16\n"
                 "v_cndmask_b32_e64 %7, %5, %7, %16\n"
                 "s_nop 0\n"
                 TSG8_QP("v_add_f32_dpp", "%3", "[1,0,3,2]") TSG8_QP("v_max_f32_dpp", "%7", "[1,0,3,2]")
                 "s_nop 1\n"
                 : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
                 : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "s"(mask_b1));
#undef TSG8_L1
#undef TSG8_L2
#undef TSG8_QP
    sum = s3;
    mx = m3;
}
__device__ __forceinline__ uint32_t row_select8(const uint32_t (&c)[8], const RowSel &r)
{
    const uint32_t s0 = r.b3 ? c[1] : c[0], s1 = r.b3 ? c[3] : c[2], s2 = r.b3 ? c[5] : c[4], s3 = r.b3 ? c[7] : c[6];
    const uint32_t t0 = r.b2 ? s1 : s0, t1 = r.b2 ? s3 : s2;
    return r.b1 ? t1 : t0;
}

// 16 values: the row-wide sum of the value fed at position (b3 + 2 b2 + 4 b1 + 8 b0) lands in the lane -- callers feed
// gradient-record column c at position bitrev4(c), so that lane (l & 15) of a group ends up with column (l & 15).
// Levels 1 and 2 pair lanes of different DPP banks (l ^ 8 via row_ror:8, then the two banks of each 8 via row_half_mirror), so
// the "which half keeps which value" select is the instruction's own bank write mask: two v_add_f32_dpp per pair instead of
// two v_cndmask + one (all half rate on gfx950).  Levels 3 and 4 pair lanes inside a quad and need one v_cndmask each.
// Written as ONE asm statement because hipcc's DPP combiner does not form partially masked adds; wait states (a VALU
// result needs 2 states before a DPP op reads it) are satisfied by the instruction order plus the three s_nop.
__device__ __forceinline__ float row_reduce16(float (&v)[16], unsigned long long mask_b1, unsigned long long mask_b0)
{
#define TSG_L1(X, Y)                                                                    \
    "v_add_f32_dpp " Y ", " Y ", " Y " row_ror:8 row_mask:0xf bank_mask:0xc\n"         \
    "v_add_f32_dpp " Y ", " X ", " X " row_ror:8 row_mask:0xf bank_mask:0x3\n"
#define TSG_L2(X, Y)                                                                    \
    "v_add_f32_dpp " Y ", " Y ", " Y " row_half_mirror row_mask:0xf bank_mask:0xa\n"   \
    "v_add_f32_dpp " Y ", " X ", " X " row_half_mirror row_mask:0xf bank_mask:0x5\n"
#define TSG_L3(X, Y, QP, M)                                                             \
    "v_add_f32_dpp " X ", " X ", " X " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"  \
    "v_add_f32_dpp " Y ", " Y ", " Y " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"  \
    "v_cndmask_b32_e64 " Y ", " X ", " Y ", " M "\n"
    asm volatile("s_nop 1\n"
                 TSG_L1("%0", "%1") TSG_L1("%2", "%3") TSG_L1("%4", "%5") TSG_L1("%6", "%7")
                 TSG_L1("%8", "%9") TSG_L1("%10", "%11") TSG_L1("%12", "%13") TSG_L1("%14", "%15")
                 TSG_L2("%1", "%3") TSG_L2("%5", "%7") TSG_L2("%9", "%11") TSG_L2("%13", "%15")
                 TSG_L3("%3", "%7", "[2,3,0,1]", "%16") TSG_L3("%11", "%15", "[2,3,0,1]", "%16")
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 0\n"
                 "v_add_f32_dpp %15, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_e64 %15, %7, %15, %17\n"
                 "s_nop 1\n"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 : "s"(mask_b1), "s"(mask_b0));
#undef TSG_L1
#undef TSG_L2
#undef TSG_L3
    return v[15];
}
// The same network when six of the sixteen columns are products  k_c(pixel) * x(pixel, entry)  of a per-pixel constant and ONE common
// per-step factor (the backward's dL/drgb and dL/dnormal: k = dL_dpixel, x = contrib).  A pair (X, Y) of such columns needs one level-1
// instruction instead of two if the two registers are filled "pre-swapped": v[X] = kA x with kA = column X's constant on lanes 0-7 of
// the row and column Y's on lanes 8-15, v[Y] = kB x with the two exchanged -- then  Y = ror8(v[Y]) + v[X]  is already the level-1 result.
// A quad of such columns (two pairs meeting at level 2) continues the idea with constants that depend on the lane's quarter and saves
// the level-2 instruction too.  Registers 0-3 = the quad, 4-5 = the pair, 6-15 as in row_reduce16: 26 DPP adds
// instead of 30.  Which lane ends up with which register is unchanged: reg(l) = 8 (l & 1) + 4 ((l >> 1) & 1) + {0, 2, 1, 3}[l >> 2].
__device__ __forceinline__ float row_reduce16c(float (&v)[16], unsigned long long mask_b1, unsigned long long mask_b0)
{
#define TSG_L1(X, Y)                                                                    \
    "v_add_f32_dpp " Y ", " Y ", " Y " row_ror:8 row_mask:0xf bank_mask:0xc\n"         \
    "v_add_f32_dpp " Y ", " X ", " X " row_ror:8 row_mask:0xf bank_mask:0x3\n"
#define TSG_L1C(X, Y) "v_add_f32_dpp " Y ", " Y ", " X " row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define TSG_L2(X, Y)                                                                    \
    "v_add_f32_dpp " Y ", " Y ", " Y " row_half_mirror row_mask:0xf bank_mask:0xa\n"   \
    "v_add_f32_dpp " Y ", " X ", " X " row_half_mirror row_mask:0xf bank_mask:0x5\n"
#define TSG_L3(X, Y, QP, M)                                                             \
    "v_add_f32_dpp " X ", " X ", " X " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"  \
    "v_add_f32_dpp " Y ", " Y ", " Y " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"  \
    "v_cndmask_b32_e64 " Y ", " X ", " Y ", " M "\n"
    asm volatile("s_nop 1\n"
                 TSG_L1C("%0", "%1") TSG_L1C("%2", "%3") TSG_L1C("%4", "%5") TSG_L1("%6", "%7")
                 TSG_L1("%8", "%9") TSG_L1("%10", "%11") TSG_L1("%12", "%13") TSG_L1("%14", "%15")
                 "v_add_f32_dpp %3, %3, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n" // level 2 of the quad: one instruction
                 TSG_L2("%5", "%7") TSG_L2("%9", "%11") TSG_L2("%13", "%15")
                 TSG_L3("%3", "%7", "[2,3,0,1]", "%16") TSG_L3("%11", "%15", "[2,3,0,1]", "%16")
                 "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 0\n"
                 "v_add_f32_dpp %15, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_e64 %15, %7, %15, %17\n"
                 "s_nop 1\n"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 : "s"(mask_b1), "s"(mask_b0));
#undef TSG_L1
#undef TSG_L1C
#undef TSG_L2
#undef TSG_L3
    return v[15];
}
constexpr int bitrev4(int c) { return ((c & 1) << 3) | ((c & 2) << 1) | ((c & 4) >> 1) | ((c & 8) >> 3); }


// ---- contrib_sum / contrib_max of a tile (forward, rich_info) ---------------------------------------------------------------
// The four quadrant waves of a tile and the four lane groups of each wave meet in two LDS arrays indexed by list position.
// LDS atomics on gfx950 (tools/lds_atomic_bench2.hip): ds_add_f32 193 cycles per wave instruction, ds_add_u32 4.9, ds_add_u64 7.0,
// ds_max_i32 4.8 -- so the sums are kept in 16.48 FIXED POINT (a (group, entry) sum is <= 16, a tile's <= 256; the smallest
// possible contribution, 1/255 * 1e-4, still carries 27 significant bits: the fixed-point sum is closer to the exact one than any
// fp32 summation order) and the maxima as the bit patterns of non-negative floats (int order == float order).
__device__ __forceinline__ unsigned long long to_fixed48(float x) // x in [0, 2^15): floor(x * 2^48), exact for x >= 2^-25
{
    const float y = x * 0x1p48f;                          // exact
    const uint32_t hi = (uint32_t)(y * 0x1p-32f);         // truncates
    const float rem = fmaf(-(float)hi, 0x1p32f, y);       // exact: in [0, 2^32)
    return ((unsigned long long)hi << 32) | (unsigned long long)(uint32_t)rem;
}
// Scattered global atomics cost one L2 line operation each (~20 G/s chip-wide, tools/atomic_scope_bench.hip).  A running maximum
// only grows, so a (possibly stale, hence smaller) plain read that already exceeds the new value proves the atomic redundant.
__device__ __forceinline__ void global_stats_add(uint32_t tid, float sm, float mx, float *contrib_sum, float *contrib_max)
{
    unsafeAtomicAdd(contrib_sum + tid, sm);
    if (mx > contrib_max[tid]) atomicMax((int *)contrib_max + tid, __float_as_int(mx));
}
template <int TCAP>
__device__ __forceinline__ void tile_stats_add(unsigned long long *tsum, int *tmax, int k, float sm, float mx, const uint32_t *tile_list,
                                               float *contrib_sum, float *contrib_max)
{
    if (k < TCAP)
    {
        __hip_atomic_fetch_add(tsum + k, to_fixed48(sm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_max(tmax + k, __float_as_int(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    else global_stats_add(tile_list[k] & TS_ID_MASK, sm, mx, contrib_sum, contrib_max); // list positions beyond the LDS arrays (a very long tile list); id bits only
}
__device__ __forceinline__ void tile_stats_flush(unsigned long long fx48, int mxbits, uint32_t tid, float *contrib_sum, float *contrib_max)
{
#if defined(TSG_PROBE) && TSG_PROBE == 8 // profiling build (results wrong): plain scattered stores in place of the two atomics -- what the ATOMIC costs
    contrib_sum[tid] = (float)((double)fx48 * 0x1p-48);
    contrib_max[tid] = __int_as_float(mxbits);
#elif defined(TSG_PROBE) && TSG_PROBE == 9 // profiling build (results wrong): ONE coalesced 8-byte store per instance at its list position
    ((float2 *)contrib_sum)[tid] = make_float2((float)((double)fx48 * 0x1p-48), __int_as_float(mxbits));
#else
    global_stats_add(tid, (float)((double)fx48 * 0x1p-48), __int_as_float(mxbits), contrib_sum, contrib_max);
#endif
}
} // namespace
