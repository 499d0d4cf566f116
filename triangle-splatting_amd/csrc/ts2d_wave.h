// ts2d_wave.h -- wave64 building blocks shared by the 2D and 3D blend kernels (gfx950): SGPR broadcast, DPP and
// v_permlane*_swap transpose-reduce networks, fast transcendental helpers, XCD-aware tile mapping.
// Costs measured on MI355X (profiles/r01_valu_microbench.txt): v_readlane 4.3 cycles, DPP add 4.5,
// v_permlane{16,32}_swap 8.3, plain VOP2 fp32 2.6, v_exp/v_rcp 8.3 per wave instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace
{
__device__ __forceinline__ float bcast(float v, int j)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
__device__ __forceinline__ uint32_t bcast(uint32_t v, int j) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); }

// Orders this wave's LDS accesses ACROSS LANES at this point of the program: the per-thread language model lets the compiler
// merge or reorder the accesses of different lanes (it did: three lane groups' read-add-write sequences became three reads and
// one common write); a wavefront-scope fence + wave_barrier pins them.  No instruction is emitted: LDS executes a wave's
// accesses in order.
__device__ __forceinline__ void wave_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

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
// Maximum of NON-NEGATIVE floats on their bit patterns: for x, y >= 0 the integer order equals the float order, and
// v_max_i32 needs none of the NaN-quieting (v_max_f32 x, x) that IEEE-mode fmaxf drags in after every cross-lane move.
struct OpMax
{
    __device__ __forceinline__ float operator()(float a, float b) const { return __int_as_float(max(__float_as_int(a), __float_as_int(b))); }
};

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

// Same idea for 8 values: each lane returns the complete reduction of ONE of the 8 values, the 8 lanes of a group
// (lane >> 3) share it.  Used by the forward's contribution statistics (8 parked entries per flush).
template <typename Op>
__device__ __forceinline__ float reduce8(float (&v)[8], int lane, Op op)
{
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        swap32(v[2 * i], v[2 * i + 1]);
        v[i] = op(v[2 * i], v[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
    {
        swap16(v[2 * i], v[2 * i + 1]);
        v[i] = op(v[2 * i], v[2 * i + 1]);
    }
    const bool b3 = lane & 8;
    const float own = b3 ? v[1] : v[0];
    const float oth = b3 ? v[0] : v[1];
    float r = op(own, dpp<DPP_ROR8>(oth));
    r = op(r, dpp<DPP_HALF_MIRROR>(r));
    r = op(r, dpp<DPP_XOR1>(r));
    r = op(r, dpp<DPP_XOR2>(r));
    return r;
}
__device__ __forceinline__ int slot8_of_lane(int lane)
{
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (lane == 0) ? (float)i : 0.0f;
    return (int)reduce8(v, lane, OpAdd());
}

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

// blockIdx -> tile.  Workgroups are handed to the 8 XCDs round-robin (block b runs on XCD b % 8) and an XCD never takes over another
// one's blocks, so the mapping decides two things: which tiles share an L2 (neighbouring tiles gather mostly the same triangle records),
// and how evenly the WORK is spread over the XCDs.
//   TS2D_XCD_BANDS (rounds 1-3): XCD x owns one contiguous band of row-major tiles (a ninth of the image height at 1080p).  Best locality,
//     but a view whose content sits in the middle of the image (an object-centric capture) leaves the XCDs of the top and bottom bands idle.
//   default (round 4): XCD x owns the tile ROWS x, x + 8, x + 16, ... of the first 8 floor(rows / 8) rows -- every XCD samples the whole
//     image height, horizontal neighbours still share an L2 (a row of 120 tiles at 1080p), vertical neighbours are fetched by two XCDs --
//     and an eighth of the tiles of the remaining rows (68 rows at 1080p: XCDs with nine rows against XCDs with eight would cost the
//     uniform scene 6 %).  Measured (profiles/r04_notes.md): the same triangles concentrated about the optical axis 1.00 instead of 1.52 ms
//     per step; the uniform headline scene within 1 % of the bands.
// Every XCD gets ceil(ntiles / 8) units; a unit past its share returns -1 (the grid is padded to 8 x that).
static inline int ts_tile_units(int grid_x, int grid_y) { return 8 * ((grid_x * grid_y + 7) / 8); }
#ifdef TS2D_XCD_BANDS
__device__ __forceinline__ int tile_of_block(int b, int grid_x, int grid_y)
{
    const int ntiles = grid_x * grid_y, q = ntiles >> 3, r = ntiles & 7, x = b & 7, i = b >> 3;
    return i < q + (x < r ? 1 : 0) ? x * q + min(x, r) + i : -1;
}
#else
__device__ __forceinline__ int tile_of_block(int b, int grid_x, int grid_y)
{
    const int x = b & 7, i = b >> 3;
    const int rows8 = grid_y >> 3, full = rows8 * grid_x; // units of an XCD that are whole rows
    if (i < full) return (x + 8 * (i / grid_x)) * grid_x + i % grid_x;
    const int rest = (grid_y & 7) * grid_x, q = rest >> 3, r = rest & 7, k = i - full; // the last grid_y % 8 rows, shared out tile by tile
    return k < q + (x < r ? 1 : 0) ? 8 * full + x * q + min(x, r) + k : -1;
}
#endif

// ---- reduce4: 4 values per lane x 64 lanes -> every lane of row r returns the complete sum of value map[r] ----
__device__ __forceinline__ float reduce4(float x0, float x1, float x2, float x3)
{
    swap32(x0, x1);
    x0 += x1;
    swap32(x2, x3);
    x2 += x3;
    swap16(x0, x2);
    float r = x0 + x2;
    r += dpp<DPP_XOR1>(r);
    r += dpp<DPP_XOR2>(r);
    r += dpp<DPP_HALF_MIRROR>(r);
    r += dpp<DPP_MIRROR>(r);
    return r;
}
__device__ __forceinline__ int slot4_of_lane(int lane)
{
    const float i = (lane == 0) ? 1.0f : 0.0f;
    return (int)reduce4(0.0f, i, 2.0f * i, 3.0f * i);
}

} // namespace
