// render_q8.hip -- per-pixel alpha blend (forward) and its back-to-front replay (backward): eight lane groups per wave, each
// walking its OWN QUEUE of triangles, decoupled from the 64-entry batches in which the tile list is culled.
//
// Behaviour follows FORWARD::renderCUDA (R2D/src/forward.cu:198-355) and BACKWARD::renderCUDA (R2D/src/backward.cu:265-493);
// the per-pixel arithmetic is that of render_group.hip (the reference's pixel-relative barycentrics).
//
// What changed against render_group.hip, and why (tools/sim/group_sim.c replays the headline scene on the CPU and reproduces the GPU
// counters of the old kernel exactly: 15.1 M (entry, block) survivors, 4.47 M wave steps, 113.4 M blended pairs):
//   * lane occupancy was 39.6 %: four 16-lane groups (4x4 pixel blocks) in lockstep, a pass costs max(list length) steps.
//     Eight 8-lane groups (4 wide x 2 high) in lockstep would take 3.95 M steps (-12 %); what the lockstep loses is the imbalance
//     between the groups' lists inside one 64-entry batch.  So the lists became QUEUES that survive the batch: a group keeps
//     consuming its queue while the wave culls the next batch only when some group runs dry and the row ring has room.  Model:
//     3.4-3.5 M steps (-22..-24 %) with the same 32 table rows.
//   * a queue entry IS the LDS byte address of the triangle's row (u16): no unpacking, no multiply per step (the old step spent
//     four half-rate instructions on that); the list position the blend needs (n_contrib, contribution statistics, the
//     backward's `k < last` test) travels in the row.
//   * rows live in a ring; a row is retired (backward: its gradient sums leave as one 64-byte atomic line operation) when no
//     queue references it any more.  The backward keeps one 64-byte sum slice per HALF of a 16-lane DPP row, so the two groups
//     of a DPP row never meet in an accumulator; groups of different DPP rows that hold the same triangle at the same step are
//     detected once per batch (their queue positions stay aligned until the next batch) and take the serialised path.
//   * gfx950 issue costs in real shader cycles (tools/valu_bench3.hip, profiles/r03_valu_microbench3.txt): fma / add / mul / and 2,
//     compare / select / min / shift / mad24 / every DPP form 4, exp / rcp 8.  The 8-lane transposed reduction below needs
//     24 DPP adds + 5 selects for the 16 gradient columns.
#include "ts2d_common.h"
#include "ts2d_wave.h"
#include "ts2d_group.h"

namespace
{
constexpr int QR = 32;               // rows of the ring (power of two)
constexpr int QMINFREE = 20;         // a batch is culled only when at least this many rows are free ...
constexpr int QLEN = 2 * QR + 8;     // ... and a queue array has room for QR live entries + QR dummies behind the tail; the 16 extra bytes
                                     // stagger the eight arrays over the LDS banks (at 128 bytes apart the groups' entries collide 4-way)
constexpr int FROWB = 80;            // forward row: 20 floats (layout below)
constexpr int BROWB = 80 + 2 * 64;   // backward row: the same 20 floats + one 16-float sum slice per half of the DPP row
constexpr int QBYTES = 8 * QLEN * 2; // eight queues of u16 row addresses
constexpr int FWAVE = (QR + 1) * FROWB + QBYTES;
constexpr int BWAVE = (QR + 1) * BROWB + QBYTES;
static_assert(FWAVE % 16 == 0 && BWAVE % 16 == 0 && (QR & (QR - 1)) == 0 && QR <= 32, "LDS carve");
// Row (floats): [0..5] u1x u1y u2x u2y u3x u3y (screen vertices relative to the quadrant origin)  [6] 1/area2  [7] opacity
//               [8..10] r g b  [11..13] normal  [14..16] vertex depths  [17] triangle id  [18] list position  [19] -
// Row QR is the dummy: a unit triangle a thousand pixels away with opacity 0 (no pixel can hit it).

typedef unsigned short __attribute__((may_alias)) u16q;

struct Cull8
{
    float u1x, u1y, u2x, u2y, u3x, u3y, ia;
    float margin[8]; // >= 0: the triangle's support can reach group g = 2 r + h (DPP row r = 4x4 block (4 (r & 1), 4 (r >> 1)), half h = its rows 2h, 2h + 1)
};

// Conservative culling of one triangle against the eight 4x2 sample blocks of the quadrant whose origin is (OX, OY); the same
// construction as block_cull of render_group.hip (support = the triangle scaled by E about its centroid, alpha >= 1/255 and
// ecc <= 10; bounding box + the three edge normals as separating axes; affine forms used HERE only, their rounding error added
// to the acceptance margin).  Every test is a "value >= 0", so a group's verdict is the minimum of seven values: no mask
// arithmetic, one compare per group.
template <bool GAMMA1>
__device__ __forceinline__ Cull8 cull8(float v1x, float v1y, float v2x, float v2y, float v3x, float v3y, float op, float g2, float OX, float OY)
{
    Cull8 s;
    const float area2 = __fsub_rn(__fmul_rn(v2x - v1x, v3y - v1y), __fmul_rn(v2y - v1y, v3x - v1x)); // forward.cu:137
    s.ia = __builtin_amdgcn_rcpf(area2);
    s.u1x = v1x - OX; s.u1y = v1y - OY; s.u2x = v2x - OX; s.u2y = v2y - OY; s.u3x = v3x - OX; s.u3y = v3y - OY;
    const float C1 = (s.u2x * s.u3y - s.u2y * s.u3x) * s.ia, A1 = (v2y - v3y) * s.ia, B1 = (v3x - v2x) * s.ia;
    const float C2 = (s.u3x * s.u1y - s.u3y * s.u1x) * s.ia, A2 = (v3y - v1y) * s.ia, B2 = (v1x - v3x) * s.ia;
    const float A3 = -A1 - A2, B3 = -B1 - B2, C3 = 1.0f - C1 - C2;
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
    const float pad = 0.05f;
    const float bminx = fmaf(E, fminf(fminf(s.u1x, s.u2x), s.u3x) - cx, cx) - pad, bmaxx = fmaf(E, fmaxf(fmaxf(s.u1x, s.u2x), s.u3x) - cx, cx) + pad;
    const float bminy = fmaf(E, fminf(fminf(s.u1y, s.u2y), s.u3y) - cy, cy) - pad, bmaxy = fmaf(E, fmaxf(fmaxf(s.u1y, s.u2y), s.u3y) - cy, cy) + pad;
    // bounding box against the block's sample positions [bx, bx + 3] x [by, by + 1]
    float xm[2], ym[4];
#pragma unroll
    for (int i = 0; i < 2; i++) xm[i] = fminf(bmaxx - (float)(4 * i), (float)(4 * i + 3) - bminx);
#pragma unroll
    for (int j = 0; j < 4; j++) ym[j] = fminf(bmaxy - (float)(2 * j), (float)(2 * j + 1) - bminy);
    if (!(E > 0.0f)) xm[0] = xm[1] = -1.0f; // alpha < 1/255 everywhere (also: a lane without an entry)
    // ecc <= E  <=>  min_k a_k >= (1 - E) / 3; the maximum of a_k over the 4x2 sample box at (bx, by) is
    // C_k + A_k bx + B_k by + max(0, 3 A_k) + max(0, B_k)
    const float m = (1.0f - E) * (1.0f / 3.0f);
    const float k1 = C1 + fmaxf(0.0f, 3.0f * A1) + fmaxf(0.0f, B1) - m + 1e-6f * (fabsf(C1) + 7.0f * (fabsf(A1) + fabsf(B1)));
    const float k2 = C2 + fmaxf(0.0f, 3.0f * A2) + fmaxf(0.0f, B2) - m + 1e-6f * (fabsf(C2) + 7.0f * (fabsf(A2) + fabsf(B2)));
    const float k3 = C3 + fmaxf(0.0f, 3.0f * A3) + fmaxf(0.0f, B3) - m + 1e-6f * (fabsf(C3) + 7.0f * (fabsf(A3) + fabsf(B3)));
    const float k1x[2] = {k1, fmaf(4.0f, A1, k1)}, k2x[2] = {k2, fmaf(4.0f, A2, k2)}, k3x[2] = {k3, fmaf(4.0f, A3, k3)};
#pragma unroll
    for (int g = 0; g < 8; g++)
    {
        const int r = g >> 1, i = r & 1, j = 2 * (r >> 1) + (g & 1);
        const float by = (float)(2 * j);
        const float e = fminf(fminf(fmaf(B1, by, k1x[i]), fmaf(B2, by, k2x[i])), fmaf(B3, by, k3x[i]));
        s.margin[g] = fminf(fminf(e, xm[i]), ym[j]);
    }
    return s;
}

// The reference's per-pixel barycentrics (forward.cu:299-305, backward.cu:383-391), see render_group.hip.
struct Bary8 { float p1x, p1y, p2x, p2y, p3x, p3y, a1, a2, a3, mn, ecc; };
__device__ __forceinline__ Bary8 barycentrics8(const float4 &q0, const float4 &q1, float fx, float fy)
{
    Bary8 b;
    b.p1x = q0.x - fx; b.p1y = q0.y - fy; b.p2x = q0.z - fx; b.p2y = q0.w - fy; b.p3x = q1.x - fx; b.p3y = q1.y - fy;
    b.a1 = (b.p2x * b.p3y - b.p2y * b.p3x) * q1.z;
    b.a2 = (b.p3x * b.p1y - b.p3y * b.p1x) * q1.z;
    b.a3 = 1.0f - b.a1 - b.a2;
    b.mn = fminf(fminf(b.a1, b.a2), b.a3);
    b.ecc = fmaf(-3.0f, b.mn, 1.0f);
    return b;
}

// ---- 8-lane transposed reductions (two groups per 16-lane DPP row) ----------------------------------------------------------
// Level A pairs lanes j, 7 - j (row_half_mirror: they sit in different DPP banks, so "which half keeps which value" is the
// instruction's bank mask: two adds per pair, no select), levels B and C pair lanes inside a quad (two adds + one select).
// Sixteen values in, two out: lane j of a group ends with the group-wide sums of inputs m(j) and 8 + m(j), m = {0,4,2,6,1,5,3,7}.
// Inputs 0..3 and 4..5 are (per-pixel constant) x (one per-step factor) columns and arrive PRE-SWAPPED (see the caller), which
// lets their level-A pairs take one add and the quad's level-B pair one add as well: 24 DPP adds + 5 selects (tools/ and
// DESIGN.md 5.2; the network was checked symbolically before it was written).
__device__ __forceinline__ void group_reduce16c(float (&v)[16], unsigned long long mask_b1, unsigned long long mask_b0)
{
#define Q8_A(X, Y)                                                                       \
    "v_add_f32_dpp " Y ", " Y ", " Y " row_half_mirror row_mask:0xf bank_mask:0xa\n"    \
    "v_add_f32_dpp " Y ", " X ", " X " row_half_mirror row_mask:0xf bank_mask:0x5\n"
#define Q8_AC(X, Y) "v_add_f32_dpp " Y ", " Y ", " X " row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define Q8_Q(X, Y, QP, M)                                                                \
    "v_add_f32_dpp " X ", " X ", " X " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"   \
    "v_add_f32_dpp " Y ", " Y ", " Y " quad_perm:" QP " row_mask:0xf bank_mask:0xf\n"   \
    "v_cndmask_b32_e64 " Y ", " X ", " Y ", " M "\n"
    asm volatile("s_nop 1\n"
                 Q8_AC("%0", "%1") Q8_AC("%2", "%3") Q8_AC("%4", "%5")
                 Q8_A("%6", "%7") Q8_A("%8", "%9") Q8_A("%10", "%11") Q8_A("%12", "%13") Q8_A("%14", "%15")
                 "v_add_f32_dpp %3, %3, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" // level B of the constant quad
                 Q8_Q("%5", "%7", "[2,3,0,1]", "%16") Q8_Q("%9", "%11", "[2,3,0,1]", "%16") Q8_Q("%13", "%15", "[2,3,0,1]", "%16")
                 Q8_Q("%3", "%7", "[1,0,3,2]", "%17") Q8_Q("%11", "%15", "[1,0,3,2]", "%17")
                 "s_nop 1\n"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                 : "s"(mask_b1), "s"(mask_b0));
#undef Q8_A
#undef Q8_AC
#undef Q8_Q
}
// Input register -> gradient-record column (0..5 screen vertices, 6 opacity, 7..9 rgb, 10..12 normal, 13..15 vertex depths):
//   inputs 0..3 = r g b nx (the constant quad), 4..5 = ny nz (the constant pair), 6..11 = the six vertex components, 12 = opacity,
//   13..15 = vertex depths.  Slice position p = 2 j + i holds input m(j) + 8 i.
__device__ __forceinline__ int q8_column_of_position(int p)
{
    // p:      0  1  2  3  4  5  6  7  8  9 10 11 12 13 14 15
    // input:  0  8  4 12  2 10  6 14  1  9  5 13  3 11  7 15
    // column: 7  2 11  6  9  4  0 14  8  3 12 13 10  5  1 15
    return (int)((0xF15ADC38E0496B27ull >> (4 * p)) & 15ull);
}

// Four values over the 8 lanes of a group (the forward's contribution statistics, one window = four steps): lanes j and j ^ 1
// end with the group-wide reduction of value (j >> 2 & 1) + 2 (j >> 1 & 1).
template <typename Op>
__device__ __forceinline__ float group_reduce4(const float (&c)[4], bool b2, bool b1, Op op)
{
    const float o0 = b2 ? c[1] : c[0], x0 = b2 ? c[0] : c[1], o1 = b2 ? c[3] : c[2], x1 = b2 ? c[2] : c[3];
    const float s0 = op(o0, dpp<DPP_HALF_MIRROR>(x0)), s1 = op(o1, dpp<DPP_HALF_MIRROR>(x1));
    const float o = b1 ? s1 : s0, x = b1 ? s0 : s1;
    const float t = op(o, dpp<DPP_XOR2>(x));
    return op(t, dpp<DPP_XOR1>(t));
}

__device__ __forceinline__ uint32_t row_slot(uint32_t rel, int rowb) // rel / rowb for row-aligned rel < 2^13
{
    return rowb == FROWB ? (rel * 52429u) >> 22 : (rel * 40330u) >> 23;
}

// Wave-uniform bookkeeping of the eight queues.
struct Queues
{
    int rem[8]; // entries still queued per group
    int pos;    // steps taken since the queues were last compacted (the same for every group: idle groups walk over dummies)
    __device__ __forceinline__ int of_group(int grp) const // lane-private copy of the own group's count (only needed around a batch)
    {
        int vec = 0; // lane g <- rem[g], then every lane fetches the lane of its group (a select chain over rem[] would be turned
                     // into an indexed load and move the counts to scratch memory)
#pragma unroll
        for (int g = 0; g < 8; g++)
        {
            const int sv = __builtin_amdgcn_readfirstlane(rem[g]);
            asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(sv), "i"(g));
        }
        return __builtin_amdgcn_ds_bpermute(grp << 2, vec);
    }
};

#ifdef TS2D_STATS
// Profiling builds only (-DTS2D_STATS), read with ts2d_stats_read_q8(): [0] list entries culled  [1] (entry, group) pairs queued
// [2] wave steps  [3] chunks  [4] (pixel, entry) pairs blended  [5] quadrant waves  [6] batches  [7] rows  [8] conflict steps  [9] partial batches
__device__ unsigned long long g_stats_q8[12];
#define TSQ_STAT(i, v) stat_acc[i] += (unsigned long long)(v)
#else
#define TSQ_STAT(i, v)
#endif

// Everything the two kernels share around one culled batch: which lanes' entries get rows, where their queue entries go.
// On entry want[g] = this lane's entry is wanted by group g and M[g] its ballot; on exit the queues are compacted and the new
// entries appended.  Returns the lanes that publish a row and the number of list entries consumed.
struct BatchPlan
{
    bool row;     // this lane owns a new row
    int consumed; // list entries of this batch that are done with (64, or fewer when the ring filled up)
};
template <int ROWB>
__device__ __forceinline__ BatchPlan plan_batch(char *smem, bool (&want)[8], unsigned long long (&M)[8], int free_rows, Queues &q, int &alloc, int lane,
                                                uint32_t wbase, uint32_t qbase, uint32_t dummy, uint32_t &qptr, int myrem, uint32_t &my_row)
{
    BatchPlan bp;
    const unsigned long long any = M[0] | M[1] | M[2] | M[3] | M[4] | M[5] | M[6] | M[7];
    bool mine = want[0] || want[1] || want[2] || want[3] || want[4] || want[5] || want[6] || want[7];
    const int rank = lane_rank(any);
    bp.consumed = 64;
    int nrows = __popcll(any);
    if (nrows > free_rows) // the ring is full: the rest of the batch is culled again later
    {
        const int cut = __builtin_ctzll(ballot(mine && rank >= free_rows));
        const unsigned long long keep = (1ull << cut) - 1ull;
        mine = mine && lane < cut;
#pragma unroll
        for (int g = 0; g < 8; g++)
        {
            M[g] &= keep;
            want[g] = want[g] && lane < cut;
        }
        bp.consumed = cut;
        nrows = free_rows;
    }
    bp.row = mine;
    const int grp = lane >> 3, j = lane & 7;
    const uint32_t myq = qbase + (uint32_t)grp * (QLEN * 2);
    // compaction: the own group's remaining entries move to the front of its array, dummies behind them
    if (q.pos != 0)
    {
        uint32_t old[QR / 8];
#pragma unroll
        for (int i = 0; i < QR / 8; i++) old[i] = (j + 8 * i < myrem) ? (uint32_t)*(const u16q *)(smem + qptr + 2 * (j + 8 * i)) : dummy;
        wave_lds_order();
        const uint32_t dd = dummy | (dummy << 16);
        *(uint4 *)(smem + myq + 16 * j) = make_uint4(dd, dd, dd, dd);
        wave_lds_order();
#pragma unroll
        for (int i = 0; i < QR / 8; i++)
            if (j + 8 * i < myrem) *(u16q *)(smem + myq + 2 * (j + 8 * i)) = (unsigned short)old[i];
        wave_lds_order();
        q.pos = 0;
    }
    qptr = myq;
    my_row = wbase + (uint32_t)((alloc + rank) & (QR - 1)) * ROWB;
#pragma unroll
    for (int g = 0; g < 8; g++)
    {
        if (want[g]) *(u16q *)(smem + qbase + g * (QLEN * 2) + 2 * (q.rem[g] + lane_rank(M[g]))) = (unsigned short)my_row;
        q.rem[g] = __builtin_amdgcn_readfirstlane(q.rem[g] + __popcll(M[g]));
    }
    alloc += nrows;
    wave_lds_order();
    return bp;
}

// How many steps until `need` rows of the ring are free?  A row is free once no queue references it; rows are allocated in ring order
// and every queue holds its entries in that order, so the rows that must retire are a prefix of each queue: count it per group (each
// lane inspects four entries of its own group's queue), take the longest.  Also returns the current age of the oldest live row.
template <int ROWB>
__device__ __forceinline__ int steps_until_free(const char *smem, uint32_t qptr, int myrem, int alloc, uint32_t wbase, int need, int &oldest_age)
{
    const int j = threadIdx.x & 7, limit = QR - need; // rows older than `limit` allocations stand in the way
    int cnt = 0, age0 = 0;
#pragma unroll
    for (int m = 0; m < QR / 8; m++)
    {
        const int i = j + 8 * m;
        if (i < myrem)
        {
            const uint32_t e = *(const u16q *)(smem + qptr + 2 * i);
            const int age = ((alloc - 1 - (int)row_slot(e - wbase, ROWB)) & (QR - 1)) + 1;
            cnt += age > limit ? 1 : 0;
            if (m == 0) age0 = age;
        }
    }
    cnt += __builtin_amdgcn_update_dpp(0, cnt, DPP_XOR1, 0xF, 0xF, false);
    cnt += __builtin_amdgcn_update_dpp(0, cnt, DPP_XOR2, 0xF, 0xF, false);
    cnt += __builtin_amdgcn_update_dpp(0, cnt, DPP_HALF_MIRROR, 0xF, 0xF, false);
    // the oldest row of a group is its first entry (lane j = 0); pack (steps, age) so that one reduction finds both maxima
    int both = (cnt << 8) | age0;
    both = max(both & 0xFF00, __builtin_amdgcn_update_dpp(0, both, DPP_ROR8, 0xF, 0xF, false) & 0xFF00) |
           max(both & 0xFF, __builtin_amdgcn_update_dpp(0, both, DPP_ROR8, 0xF, 0xF, false) & 0xFF);
    int steps = 0;
    oldest_age = 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
    {
        // lanes 0 (bank 0: entry 0 of group 2r) and 8 (group 2r + 1) were merged by the row_ror:8 step above
        const int x = __builtin_amdgcn_readlane(both, 16 * r);
        steps = max(steps, x >> 8);
        oldest_age = max(oldest_age, x & 0xFF);
    }
    return steps;
}

// Age (in allocations) of the oldest row some queue still references: each queue's first entry.
template <int ROWB>
__device__ __forceinline__ int ring_age(const char *smem, uint32_t qptr, int myrem, int alloc, uint32_t wbase)
{
    int age = 0;
    if (myrem > 0) age = ((alloc - 1 - (int)row_slot((uint32_t)*(const u16q *)(smem + qptr) - wbase, ROWB)) & (QR - 1)) + 1;
    age = max(age, __builtin_amdgcn_update_dpp(0, age, DPP_ROR8, 0xF, 0xF, false)); // the two groups of a DPP row
    return max(max(__builtin_amdgcn_readlane(age, 0), __builtin_amdgcn_readlane(age, 16)),
               max(__builtin_amdgcn_readlane(age, 32), __builtin_amdgcn_readlane(age, 48)));
}

__device__ __forceinline__ void q8_init_wave(char *smem, int lane, uint32_t qbase, uint32_t dummy, int rowb)
{
    if (lane < 20)
    {
        float v = 0.0f;
        if (lane == 0 || lane == 1 || lane == 3 || lane == 4) v = 1000.0f;
        if (lane == 2 || lane == 5) v = 1001.0f;
        if (lane == 6) v = 1.0f;
        *(float *)(smem + dummy + 4 * lane) = v;
    }
    const uint32_t dd = dummy | (dummy << 16);
    for (int i = lane; i < QBYTES / 16; i += 64) *(uint4 *)(smem + qbase + 16 * i) = make_uint4(dd, dd, dd, dd); // the eight queue arrays
    (void)rowb;
}

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256, 6) render_fwd_q8_kernel(RenderArgs a, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
                                                             const float4 *__restrict__ rec, float *__restrict__ final_T,
                                                             uint32_t *__restrict__ n_contrib, float *__restrict__ out_feature,
                                                             float *__restrict__ out_depth, float *__restrict__ out_normal,
                                                             float *__restrict__ contrib_sum, float *__restrict__ contrib_max)
{
    __shared__ __attribute__((aligned(16))) char smem[4 * FWAVE];
    // contrib_sum / contrib_max of the tile's first TCAP list entries, merged over the four quadrant waves before they leave
    // as global atomics (ts2d_group.h)
    constexpr int TCAP = 896;
    __shared__ unsigned long long tsum[RICH ? TCAP : 1]; // 16.48 fixed point
    __shared__ int tmax[RICH ? TCAP : 1];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane >> 4, sub = lane & 15;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = ((row & 1) << 2) + (sub & 3), ly = ((row >> 1) << 2) + (sub >> 2);
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
    const uint32_t wbase = (uint32_t)wave * FWAVE, dummy = wbase + QR * FROWB, qbase = wbase + (QR + 1) * FROWB;
    q8_init_wave(smem, lane, qbase, dummy, FROWB);
    wave_lds_order();
    const bool b2 = lane & 4, b1 = lane & 2;
    const int wstep = (b2 ? 1 : 0) + (b1 ? 2 : 0); // which of a window's four steps this lane reports

    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, anx = 0.0f, any_ = 0.0f, anz = 0.0f, ad = 0.0f;
    bool done = !inside;
    uint32_t last = (uint32_t)len; // a pixel that never saturates examines the whole list (forward.cu:296-297)

    Queues q;
#pragma unroll
    for (int g = 0; g < 8; g++) q.rem[g] = 0;
    q.pos = 0;
    int cursor = 0, alloc = 0;
    uint32_t qptr = qbase + (uint32_t)(lane >> 3) * (QLEN * 2);
#ifdef TS2D_STATS
    unsigned long long stat_acc[12] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
#endif

    bool ready = false; // the next batch can be culled right away: a group with live pixels is dry and the ring has room
    for (;;)
    {
        const unsigned long long nd = ballot(!done);
        if (nd == 0) break;
        int t_dry = 1 << 20, kmax = 0;
#pragma unroll
        for (int g = 0; g < 8; g++)
        {
            const bool alive = ((nd >> (8 * g)) & 0xFFull) != 0;
            if (!alive) q.rem[g] = 0; // a group whose pixels are all saturated drops what it had queued
        }
        if (cursor < len && ready)
        {
            // ---- cull the next batch of 64 list entries, one per lane ----
            const int myrem = q.of_group(lane >> 3);
            const int free_rows = QR - ring_age<FROWB>(smem, qptr, myrem, alloc, wbase);
            const int k = cursor + lane;
            const bool valid = k < len;
            uint32_t id = 0;
            float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
            if (valid)
            {
                id = point_list[range.x + k] & TS_ID_MASK; // id bits (ts2d_support.h)
                const float4 *rp = rec + 4 * (size_t)id;
                r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
                if (RICH) r3 = rp[3];
            }
            const Cull8 s = cull8<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY); // a lane without an entry: opacity 0, no group
            bool want[8];
            unsigned long long M[8];
#pragma unroll
            for (int g = 0; g < 8; g++)
            {
                want[g] = s.margin[g] >= 0.0f && ((nd >> (8 * g)) & 0xFFull) != 0;
                M[g] = ballot(want[g]);
            }
            TSQ_STAT(6, 1);
            uint32_t my_row;
            const BatchPlan bp = plan_batch<FROWB>(smem, want, M, free_rows, q, alloc, lane, wbase, qbase, dummy, qptr, myrem, my_row);
            TSQ_STAT(0, min(bp.consumed, len - cursor));
            TSQ_STAT(7, __popcll(ballot(bp.row)));
            TSQ_STAT(9, bp.consumed < 64 ? 1 : 0);
#ifdef TS2D_STATS
            for (int g = 0; g < 8; g++) TSQ_STAT(1, __popcll(M[g]));
#endif
            if (bp.row)
            {
                float4 *w = (float4 *)(smem + my_row);
                w[0] = make_float4(s.u1x, s.u1y, s.u2x, s.u2y);
                w[1] = make_float4(s.u3x, s.u3y, s.ia, r1.z);
                w[2] = make_float4(r1.w, r2.x, r2.y, r2.z);
                w[3] = make_float4(r2.w, r3.x, r3.y, r3.z);
                w[4] = make_float4(r3.w, __uint_as_float(id), __uint_as_float((uint32_t)k), 0.0f);
            }
            wave_lds_order();
            cursor += bp.consumed;
            ready = false;
        }
        // How many steps until the next batch can be culled?  Until some group with live pixels has run dry (t_dry) AND the ring has
        // QMINFREE free rows (t_room): exactly one chunk of steps per batch.
#pragma unroll
        for (int g = 0; g < 8; g++)
        {
            if (((nd >> (8 * g)) & 0xFFull) != 0) t_dry = min(t_dry, q.rem[g]);
            kmax = max(kmax, q.rem[g]);
        }
        int kmin = kmax; // list exhausted: drain the queues
        if (cursor < len)
        {
            int age;
            const int t_room = steps_until_free<FROWB>(smem, qptr, q.of_group(lane >> 3), alloc, wbase, QMINFREE, age);
            kmin = __builtin_amdgcn_readfirstlane(min(max(t_dry, t_room), kmax));
            ready = true;
            if (kmin == 0) continue;
        }
        if (kmin == 0) break; // nothing queued and nothing left to cull
        TSQ_STAT(2, kmin);
        TSQ_STAT(3, 1);

        // ---- kmin steps: every group blends its next kmin queue entries (an idle group walks over dummies) ----
        for (int t0 = 0; t0 < kmin; t0 += 4)
        {
            if (t0 > 0 && ballot(!done) == 0) break; // every pixel of the quadrant is saturated
            float c[4];
#pragma unroll
            for (int st = 0; st < 4; st++)
            {
                c[st] = 0.0f;
                if (t0 + st < kmin)
                {
                    const uint32_t ra = *(const u16q *)(smem + qptr + 2 * (t0 + st));
                    const float4 q0 = *(const float4 *)(smem + ra), q1 = *(const float4 *)(smem + ra + 16);
                    const Bary8 b = barycentrics8(q0, q1, fx, fy);
                    const float4 q2 = *(const float4 *)(smem + ra + 32);
                    float4 q3 = make_float4(0, 0, 0, 0);
                    const float4 q4 = *(const float4 *)(smem + ra + 64);
                    if (RICH) q3 = *(const float4 *)(smem + ra + 48);
                    const float pw = GAMMA1 ? b.ecc * b.ecc : pow_nonneg(b.ecc, g2);
                    const float alpha = fminf(0.99f, q1.w * __builtin_amdgcn_exp2f(pw * -0.7213475204444817f)); // forward.cu:311-312
                    const bool hit = !done && ecc_in_range(b.ecc) && alpha >= 1.0f / 255.0f;                     // forward.cu:307,313
                    const float al = hit ? alpha : 0.0f; // branch-free: x + c * 0 == x, T * 1 == T bit for bit
                    TSQ_STAT(4, __popcll(ballot(hit)));
                    const float contrib = al * T;
                    ar = fmaf(q2.x, contrib, ar);
                    ag = fmaf(q2.y, contrib, ag);
                    ab = fmaf(q2.z, contrib, ab);
                    if (RICH)
                    {
                        anx = fmaf(q2.w, contrib, anx);
                        any_ = fmaf(q3.x, contrib, any_);
                        anz = fmaf(q3.y, contrib, anz);
                        const float d = q3.z * b.a1 + q3.w * b.a2 + q4.x * b.a3; // forward.cu:328
                        ad = fmaf(d, contrib, ad);
                        c[st] = contrib;
                    }
                    T *= (1.0f - al);
                    const bool sat = hit && T <= 0.0001f; // forward.cu:333
                    last = sat ? __float_as_uint(q4.z) + 1u : last;
                    done = done || sat;
                }
            }
            if (RICH)
            {
                // contrib_sum / contrib_max (forward.cu:323-324): the window's 4 x 64 contributions are reduced inside each 8-lane
                // group; lanes (j, j ^ 1) end with (sum, max) of step `wstep` of their group and the even lanes add them to the TILE's
                // statistics in LDS with integer atomics (ts2d_group.h).  The list position comes back from the row.
                const float sm = group_reduce4(c, b2, b1, OpAdd());
                const float mx = group_reduce4(c, b2, b1, OpMax());
                if ((lane & 1) == 0 && sm > 0.0f)
                {
                    const uint32_t ra = *(const u16q *)(smem + qptr + 2 * (t0 + wstep));
                    const int k = (int)*(const uint32_t *)(smem + ra + 72);
                    tile_stats_add<TCAP>(tsum, tmax, k, sm, mx, point_list + range.x, contrib_sum, contrib_max);
                }
            }
        }
        qptr += 2 * kmin;
        q.pos += kmin;
#pragma unroll
        for (int g = 0; g < 8; g++) q.rem[g] = __builtin_amdgcn_readfirstlane(max(q.rem[g] - kmin, 0));
    }

#ifdef TS2D_STATS
    if (lane == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&g_stats_q8[i], stat_acc[i]);
#endif
    if (RICH)
    {
        __syncthreads(); // the only rendezvous of the four quadrant waves: the tile's merged contribution statistics leave
        for (int k = threadIdx.x; k < min(len, TCAP); k += 256)
        {
            const unsigned long long fx48 = tsum[k];
            if (fx48 != 0ull) tile_stats_flush(fx48, tmax[k], point_list[range.x + k] & TS_ID_MASK, contrib_sum, contrib_max);
        }
    }
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

// Backward.  Per (pixel, triangle) pair the reference adds 16 values into per-triangle arrays (backward.cu:412-490); the pair's
// 16 values are formed per lane exactly as in render_group.hip (the reference's per-pixel form; the division by area2 is applied
// once per triangle when its sums leave), reduced over the group's 8 lanes and added to the row's sum slice of this half.
template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256, 5) render_bwd_q8_kernel(RenderArgs a, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list,
                                                             const float4 *__restrict__ rec, const float *__restrict__ final_T,
                                                             const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dout_feature,
                                                             const float *__restrict__ dL_dout_depth, const float *__restrict__ dL_dout_normal,
                                                             float *__restrict__ grad_rec)
{
    __shared__ __attribute__((aligned(16))) char smem[4 * BWAVE];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane >> 4, sub = lane & 15;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = ((row & 1) << 2) + (sub & 3), ly = ((row >> 1) << 2) + (sub >> 2);
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)lx, fy = (float)ly, OX = (float)X0, OY = (float)Y0;
    const uint2 range = ranges[tile];
    const float g2 = 2.0f * a.gamma;
    const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
    const uint32_t wbase = (uint32_t)wave * BWAVE, dummy = wbase + QR * BROWB, qbase = wbase + (QR + 1) * BROWB;
    q8_init_wave(smem, lane, qbase, dummy, BROWB);
    {   // every sum slice starts at zero and is zeroed again by whoever flushes it
        for (int i = lane; i < (QR + 1) * 32; i += 64) *(float *)(smem + wbase + (uint32_t)(i >> 5) * BROWB + 80 + 4 * (i & 31)) = 0.0f;
    }
    wave_lds_order();

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
    // Pre-swapped constant columns of group_reduce16c: with T(x, y) = x + 2 y over (r, g, b, nx) and this lane's bits b2 = lane & 4,
    // b1 = lane & 2, input 0 carries column T(b2, b1), input 1 T(!b2, !b1), input 2 T(b2, !b1), input 3 T(!b2, b1); the pair
    // (ny, nz) is ordered by b2.
    const bool b2 = lane & 4, b1 = lane & 2;
    const float kq0 = b1 ? (b2 ? dnx : dpb) : (b2 ? dpg : dpr);
    const float kq1 = b1 ? (b2 ? dpr : dpg) : (b2 ? dpb : dnx);
    const float kq2 = b1 ? (b2 ? dpg : dpr) : (b2 ? dnx : dpb);
    const float kq3 = b1 ? (b2 ? dpb : dnx) : (b2 ? dpr : dpg);
    const float kp4 = b2 ? dnz : dny, kp5 = b2 ? dny : dnz;
    const uint32_t accoff = 80u + 64u * ((lane >> 3) & 1) + 8u * (lane & 7); // this lane's two sums inside a row

    // entries at list positions >= the largest n_contrib of a group are skipped by all of its pixels (backward.cu:377-379)
    float lm = (float)last;
    lm = fmaxf(lm, dpp<DPP_XOR1>(lm));
    lm = fmaxf(lm, dpp<DPP_XOR2>(lm));
    lm = fmaxf(lm, dpp<DPP_HALF_MIRROR>(lm));
    int glast[8];
#pragma unroll
    for (int g = 0; g < 8; g++) glast[g] = (int)__builtin_amdgcn_readlane((int)lm, 8 * g);
    int maxlast = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) maxlast = max(maxlast, glast[g]);
    if (maxlast <= 0) return;

    Queues q;
#pragma unroll
    for (int g = 0; g < 8; g++) q.rem[g] = 0;
    q.pos = 0;
    int cursor = maxlast; // entries [0, cursor) are still to be culled; back to front: lane l of a batch takes position cursor - 1 - l
    int alloc = 0, flushed = 0;
    uint32_t qptr = qbase + (uint32_t)(lane >> 3) * (QLEN * 2);
    unsigned long long conflict = 0;
#ifdef TS2D_STATS
    unsigned long long stat_acc[12] = {0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0};
#endif

    // rows [flushed, upto) of the ring leave: 16 lanes add the 16 floats (one 64-byte line) of one triangle's gradient record,
    // four rows per instruction; the vertex columns get their 1 / area2 here
    auto flush_rows = [&](int upto) {
        const int p = lane & 15, col = q8_column_of_position(p);
#pragma unroll 1
        for (int s0 = flushed; s0 < upto; s0 += 4)
        {
            const int s = s0 + (lane >> 4);
            if (s < upto)
            {
                const uint32_t ro = wbase + (uint32_t)(s & (QR - 1)) * BROWB;
                float *s0p = (float *)(smem + ro + 80 + 4 * p), *s1p = (float *)(smem + ro + 144 + 4 * p);
                float val = *s0p + *s1p;
                *s0p = 0.0f;
                *s1p = 0.0f;
                const uint32_t eid = __float_as_uint(*(const float *)(smem + ro + 68));
                if (col < 6) val *= *(const float *)(smem + ro + 24);
                if (RICH || col < 10) unsafeAtomicAdd(grad_rec + TS_GRAD_FLOATS * (size_t)eid + col, val);
            }
        }
        flushed = upto;
        wave_lds_order();
    };

    bool ready = false; // the next batch can be culled right away
    for (;;)
    {
        if (cursor > 0 && ready)
        {
            const int myrem = q.of_group(lane >> 3);
            const int age = ring_age<BROWB>(smem, qptr, myrem, alloc, wbase);
            flush_rows(alloc - age);
            // ---- cull the next batch, back to front ----
            const int k = cursor - 1 - lane;
            const bool valid = k >= 0;
            uint32_t id = 0;
            float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
            if (valid)
            {
                id = point_list[range.x + k] & TS_ID_MASK; // id bits (ts2d_support.h)
                const float4 *rp = rec + 4 * (size_t)id;
                r0 = rp[0]; r1 = rp[1]; r2 = rp[2];
                if (RICH) r3 = rp[3];
            }
            const Cull8 s = cull8<GAMMA1>(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, g2, OX, OY); // a lane without an entry: opacity 0, no group
            bool want[8];
            unsigned long long M[8];
#pragma unroll
            for (int g = 0; g < 8; g++)
            {
                want[g] = s.margin[g] >= 0.0f && k < glast[g];
                M[g] = ballot(want[g]);
            }
            TSQ_STAT(6, 1);
            uint32_t my_row;
            const BatchPlan bp = plan_batch<BROWB>(smem, want, M, QR - age, q, alloc, lane, wbase, qbase, dummy, qptr, myrem, my_row);
            TSQ_STAT(7, __popcll(ballot(bp.row)));
            TSQ_STAT(9, bp.consumed < 64 ? 1 : 0);
#ifdef TS2D_STATS
            for (int g = 0; g < 8; g++) TSQ_STAT(1, __popcll(M[g]));
#endif
            if (bp.row)
            {
                float4 *w = (float4 *)(smem + my_row);
                w[0] = make_float4(s.u1x, s.u1y, s.u2x, s.u2y);
                w[1] = make_float4(s.u3x, s.u3y, s.ia, r1.z);
                w[2] = make_float4(r1.w, r2.x, r2.y, r2.z);
                w[3] = make_float4(r2.w, r3.x, r3.y, r3.z);
                w[4] = make_float4(r3.w, __uint_as_float(id), __uint_as_float((uint32_t)k), 0.0f);
            }
            wave_lds_order();
            cursor -= bp.consumed;
            ready = false;
            // steps at which two groups of the SAME half (different DPP rows) hold the same row: they share its sum slice, so their
            // sums are added one after the other.  All queues advance together, hence one mask per batch.
            {
                const uint32_t qa = qbase + 2 * (uint32_t)(lane & (QR - 1));
                uint32_t e[8];
#pragma unroll
                for (int g = 0; g < 8; g++) e[g] = *(const u16q *)(smem + qa + g * (QLEN * 2));
                bool cf = false;
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    const uint32_t x0 = e[h], x1 = e[2 + h], x2 = e[4 + h], x3 = e[6 + h];
                    cf = cf || (x0 != dummy && (x0 == x1 || x0 == x2 || x0 == x3)) || (x1 != dummy && (x1 == x2 || x1 == x3)) || (x2 != dummy && x2 == x3);
                }
                conflict = ballot(cf && lane < QR);
            }
        }
        // one chunk of steps per batch, as in the forward: the next batch is culled when a group that still wants entries has run dry
        // and the ring has room
        int t_dry = 1 << 20, kmax = 0;
#pragma unroll
        for (int g = 0; g < 8; g++)
        {
            if (min(cursor, glast[g]) > 0) t_dry = min(t_dry, q.rem[g]);
            kmax = max(kmax, q.rem[g]);
        }
        int kmin = kmax; // list exhausted: drain the queues
        if (cursor > 0)
        {
            int age;
            const int t_room = steps_until_free<BROWB>(smem, qptr, q.of_group(lane >> 3), alloc, wbase, QMINFREE, age);
            kmin = __builtin_amdgcn_readfirstlane(min(max(t_dry, t_room), kmax));
            ready = true;
            if (kmin == 0) continue;
        }
        if (kmin == 0) break; // nothing queued, nothing left to cull
        TSQ_STAT(2, kmin);
        TSQ_STAT(3, 1);
        TSQ_STAT(8, __popcll((conflict >> q.pos) & ((1ull << kmin) - 1ull)));

        for (int t = 0; t < kmin; t++)
        {
            const uint32_t ra = *(const u16q *)(smem + qptr + 2 * t);
            float2 *acc = (float2 *)(smem + ra + accoff);
            const bool shared_row = (bool)((conflict >> (q.pos + t)) & 1); // wave-uniform
            const float2 acc0 = *acc;                                      // fetched early; only used when no other group adds to this slice now
            const float4 q0 = *(const float4 *)(smem + ra), q1 = *(const float4 *)(smem + ra + 16);
            const Bary8 b = barycentrics8(q0, q1, fx, fy);
            const float4 q2 = *(const float4 *)(smem + ra + 32);
            float4 q3 = make_float4(0, 0, 0, 0);
            const float4 q4 = *(const float4 *)(smem + ra + 64);
            if (RICH) q3 = *(const float4 *)(smem + ra + 48);
            const float vd3 = RICH ? q4.x : 0.0f;
            const float pw = GAMMA1 ? b.ecc * b.ecc : pow_nonneg(b.ecc, g2);
            const float G = __builtin_amdgcn_exp2f(pw * -0.7213475204444817f); // exp(-0.5 pw)
            const float opG = q1.w * G;
            const float alpha = fminf(0.99f, opG);
            const bool hit = ((int)__float_as_uint(q4.z) < last) && ecc_in_range(b.ecc) && alpha >= 1.0f / 255.0f; // backward.cu:378,393,400
            TSQ_STAT(4, __popcll(ballot(hit)));
            // branch-free from here on: a lane that does not hit runs with alpha = 0, so T and B stay bit-unchanged and every value it
            // feeds into the reduction is an exact 0
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
            // backward.cu:443-447: dL_decc = dL_dpower * 2 gamma * power / (ecc + 1e-8) with power = -0.5 pw and dL_dpower = dL_dalpha * alpha
            // unless the 0.99 clamp was active; z = -3 dL_decc goes to the arg-min barycentric.  For gamma = 1, pw / (ecc + 1e-8) is ecc
            // to 1e-8 / ecc relative (a pair with ecc that small contributes ~ecc to begin with): one multiplication instead of a
            // reciprocal.
            const float zr = GAMMA1 ? 1.5f * g2 * (dL_dalpha * alpha) * b.ecc
                                    : 1.5f * g2 * (dL_dalpha * alpha) * pw * __builtin_amdgcn_rcpf(b.ecc + 1e-8f);
            const float z = (hit && opG < 0.99f) ? zr : 0.0f; // the select sits last: a lane that does not hit may hold inf / NaN in pw
            const bool k1 = b.a1 == b.mn;                     // backward.cu:449-461: a1 <= a2 && a1 <= a3, then a2 <= a1 && a2 <= a3, else a3
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
            group_reduce16c(v, 0xCCCCCCCCCCCCCCCCull, 0xAAAAAAAAAAAAAAAAull);
            if (!shared_row) *acc = make_float2(acc0.x + v[7], acc0.y + v[15]);
            else
            {
#pragma unroll
                for (int r = 0; r < 4; r++)
                {
                    if (row == r)
                    {
                        const float2 o = *acc;
                        *acc = make_float2(o.x + v[7], o.y + v[15]);
                    }
                    wave_lds_order();
                }
            }
        }
        qptr += 2 * kmin;
        q.pos += kmin;
#pragma unroll
        for (int g = 0; g < 8; g++) q.rem[g] = __builtin_amdgcn_readfirstlane(max(q.rem[g] - kmin, 0));
    }
    flush_rows(alloc);
#ifdef TS2D_STATS
    if (lane == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&g_stats_q8[i], stat_acc[i]);
#endif
}
} // namespace

#define TS_DISPATCH_Q8(KERNEL, ...)                                                                                   \
    do                                                                                                                \
    {                                                                                                                 \
        const bool g1 = (a.gamma == 1.0f);                                                                            \
        if (a.rich_info && g1) hipLaunchKernelGGL((KERNEL<true, true>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (a.rich_info) hipLaunchKernelGGL((KERNEL<true, false>), grid, dim3(256), 0, s, __VA_ARGS__);          \
        else if (g1) hipLaunchKernelGGL((KERNEL<false, true>), grid, dim3(256), 0, s, __VA_ARGS__);                   \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, dim3(256), 0, s, __VA_ARGS__);                          \
    } while (0)

void ts_launch_render_fwd_q8(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                             float *out_feature, float *out_depth, float *out_normal, float *contrib_sum, float *contrib_max, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y ? ts_tile_units(a.grid_x, a.grid_y) : 0));
    if (grid.x == 0) return;
    TS_DISPATCH_Q8(render_fwd_q8_kernel, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature, out_depth, out_normal, contrib_sum,
                   contrib_max);
}

void ts_launch_render_bwd_q8(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                             const float *dL_dout_feature, const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y ? ts_tile_units(a.grid_x, a.grid_y) : 0));
    if (grid.x == 0) return;
    TS_DISPATCH_Q8(render_bwd_q8_kernel, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, dL_dout_feature, dL_dout_depth, dL_dout_normal,
                   grad_rec);
}

#ifdef TS2D_STATS
extern "C" __attribute__((visibility("default"))) int ts2d_stats_read_q8(unsigned long long *out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stats_q8), sizeof(unsigned long long) * 12);
    if (e == hipSuccess && reset)
    {
        unsigned long long z[12] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stats_q8), z, sizeof(z));
    }
    return e == hipSuccess ? 0 : 2;
}
#endif
