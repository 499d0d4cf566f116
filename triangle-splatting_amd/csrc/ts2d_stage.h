// ts2d_stage.h -- wave-cooperative staging of per-triangle rows through LDS.
//
// One lane per triangle means each lane wants a private row of 9 (vertex) or 3M (SH) floats: read straight from HBM that
// is a 36- / 192-byte stride between lanes, every load instruction touches 64 different cache lines, and with ~10 waves
// per CU the lines are evicted from the vector L1 before the next instruction of the same wave comes back for their
// next 16 bytes.  Instead a single-wave workgroup copies its 64 rows as ONE contiguous block with dwordx4 loads
// (consecutive lanes = consecutive 16-byte pieces), parks them in LDS with an odd row stride (bank-conflict-free row
// reads), and each lane then reads only its own row.  Outputs take the same road in reverse.
// Requires 16-byte aligned base pointers (the launchers check and fall back to the direct kernels otherwise).
#pragma once
#include <hip/hip_runtime.h>

namespace ts
{
// rows [row0, row0 + 64) of a row-major (nrows x ROW) float matrix -> lds[r * STRIDE + c]
template <int ROW, int STRIDE>
__device__ __forceinline__ void stage_rows_in(float *lds, const float *__restrict__ src, int row0, int nrows, int lane)
{
    const int total = min(64, nrows - row0) * ROW;
    const float *base = src + (size_t)row0 * ROW;
    constexpr int ITERS = (64 * ROW / 4 + 63) / 64;
#pragma unroll
    for (int it = 0; it < ITERS; it++)
    {
        const int i = (it * 64 + lane) * 4;
        if (i >= 64 * ROW) break;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (i + 3 < total)
        {
            const float4 q = *(const float4 *)(base + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
        else
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < total) v[k] = base[i + k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int e = i + k;
            lds[(e / ROW) * STRIDE + (e % ROW)] = v[k];
        }
    }
}

// lds[r * STRIDE + c] -> rows [row0, row0 + 64) of a row-major (nrows x ROW) float matrix
template <int ROW, int STRIDE>
__device__ __forceinline__ void stage_rows_out(const float *lds, float *__restrict__ dst, int row0, int nrows, int lane)
{
    const int total = min(64, nrows - row0) * ROW;
    float *base = dst + (size_t)row0 * ROW;
    constexpr int ITERS = (64 * ROW / 4 + 63) / 64;
#pragma unroll
    for (int it = 0; it < ITERS; it++)
    {
        const int i = (it * 64 + lane) * 4;
        if (i >= 64 * ROW) break;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int e = i + k;
            v[k] = lds[(e / ROW) * STRIDE + (e % ROW)];
        }
        if (i + 3 < total) *(float4 *)(base + i) = make_float4(v[0], v[1], v[2], v[3]);
        else
        {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < total) base[i + k] = v[k];
        }
    }
}

static inline bool aligned16(const void *p) { return ((size_t)p & 15) == 0; }
} // namespace ts
