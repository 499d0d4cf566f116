// binning.hip -- instance emission, (tile, depth) sort and tile ranges.
//
// Follows duplicateWithKeys / identifyTileRanges and the two CUB calls of the reference
// (R2D/src/rasterizer.cu:37-75, 79-99, 186, 211-218).  Integer-exact: the sorted instance list must equal the
// reference's (stable sort on the (tile << 32 | depth bits) key, ties in ascending triangle id).
//
// Round-1 note: prefix sum and radix sort go through rocPRIM (AMD's native device primitives; its radix sort is
// the onesweep LDS-histogram design tuned per gfx target).  DESIGN.md lists the structured replacement
// (depth-sort P keys once, then a stable 1-pass multisplit over tiles) as the next step for this row.
#include "ts2d_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace
{
// One wave per 64 triangles.  Triangles covering up to SMALL tiles are emitted by their own lane; larger ones
// (the stress scenes where a triangle spans thousands of tiles) are emitted cooperatively by the whole wave so
// that a single lane never serialises a long loop.  Output order is identical to the reference's per-thread
// loop: triangle-major, then row-major tiles (rasterizer.cu:63-73).
constexpr uint32_t SMALL = 32;

__global__ void __launch_bounds__(256) emit_keys_kernel(int P, int grid_x, GeometryStateView g, BinningStateView b)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = idx < P;
    const uint32_t tiles = valid ? g.tiles_touched[idx] : 0u;
    uint2 rect = {0u, 0u};
    uint32_t dbits = 0, off = 0;
    if (tiles > 0)
    {
        rect = g.rect[idx];
        dbits = __float_as_uint(g.depth[idx]);
        off = g.offsets[idx] - tiles; // == offsets[idx-1] (inclusive scan), rasterizer.cu:57
    }
    const uint32_t minx = rect.x & 0xffffu, miny = rect.x >> 16, maxx = rect.y & 0xffffu, maxy = rect.y >> 16;
    if (tiles > 0 && tiles <= SMALL)
    {
        uint32_t o = off;
        for (uint32_t y = miny; y < maxy; y++)
            for (uint32_t x = minx; x < maxx; x++)
            {
                b.keys_unsorted[o] = ((uint64_t)(y * grid_x + x) << 32) | dbits;
                b.vals_unsorted[o] = (uint32_t)idx;
                o++;
            }
    }
    unsigned long long big = __ballot(tiles > SMALL);
    while (big)
    {
        const int j = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t t_minx = __shfl(minx, j), t_miny = __shfl(miny, j), t_maxx = __shfl(maxx, j);
        const uint32_t t_tiles = __shfl(tiles, j), t_off = __shfl(off, j), t_dbits = __shfl(dbits, j);
        const uint32_t t_idx = (uint32_t)(idx - lane + j);
        const uint32_t w = t_maxx - t_minx;
        for (uint32_t i = lane; i < t_tiles; i += 64)
        {
            const uint32_t y = t_miny + i / w, x = t_minx + i % w;
            b.keys_unsorted[t_off + i] = ((uint64_t)(y * grid_x + x) << 32) | t_dbits;
            b.vals_unsorted[t_off + i] = t_idx;
        }
    }
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t N, const uint64_t *__restrict__ keys, uint2 *__restrict__ ranges)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t cur = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[cur].x = 0;
    else
    {
        const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
        if (cur != prev)
        {
            ranges[prev].y = (uint32_t)i;
            ranges[cur].x = (uint32_t)i;
        }
    }
    if (i == N - 1) ranges[cur].y = (uint32_t)N;
}
} // namespace

size_t ts_scan_temp_bytes(int32_t P)
{
    size_t bytes = 0;
    if (P <= 0) return 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)P, rocprim::plus<uint32_t>());
    return bytes;
}

size_t ts_sort_temp_bytes(int64_t N, int end_bit)
{
    size_t bytes = 0;
    if (N <= 0) return 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)N, 0u, (unsigned)end_bit);
    return bytes;
}

hipError_t ts_scan_offsets(const GeometryStateView &g, int32_t P, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    size_t bytes = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_touched, g.offsets, (size_t)P, rocprim::plus<uint32_t>(), s);
}

void ts_launch_emit_keys(int P, int grid_x, const GeometryStateView &g, const BinningStateView &b, hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(emit_keys_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, grid_x, g, b);
}

hipError_t ts_sort_pairs(const BinningStateView &b, int64_t N, int end_bit, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    size_t bytes = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs(b.sort_temp, bytes, b.keys_unsorted, b.keys, b.vals_unsorted, b.vals, (size_t)N, 0u,
                                     (unsigned)end_bit, s);
}

void ts_launch_tile_ranges(int64_t N, const BinningStateView &b, const ImageStateView &im, hipStream_t s)
{
    if (N <= 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, b.keys, im.ranges);
}
