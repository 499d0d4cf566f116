// binning.hip -- depth ordering, instance emission, tile grouping and tile ranges.
//
// Result contract (integer-exact, what the blend kernels and the parity tests rely on): the instance list is ordered
// by (tile id, depth bit pattern, triangle id) -- exactly what the reference obtains with duplicateWithKeys +
// one stable cub::DeviceRadixSort::SortPairs over N 64-bit (tile << 32 | depth) keys + identifyTileRanges
// (R2D/src/rasterizer.cu:37-75, 210-218, 79-99).
//
// How it is obtained here (same order, ~4.5x less sort traffic; N ~ 4.6 x P for the headline scene):
//   1. stable radix sort of the P triangles by their 32-bit depth key (values = ascending ids): 4 passes x P pairs;
//   2. gather tiles_touched in that order + inclusive scan -> instance slots of the i-th nearest triangle;
//   3. emit (tile, id) instances in depth order;
//   4. stable radix sort of the N instances by TILE ID ONLY (13 bits at 1080p -> 2 passes x N x 8 B instead of
//      6 passes x N x 12 B); stability keeps the depth order (and the id order among equal depths) inside a tile;
//   5. tile ranges from the sorted tile ids.
// Sorting and scanning go through rocPRIM (AMD's native device primitives; its radix sort is the LDS-histogram
// onesweep design tuned per gfx target); the emission / gather / range kernels are ours.
#include "ts2d_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace
{
__global__ void __launch_bounds__(256) gather_tiles_kernel(int P, const uint32_t *__restrict__ perm,
                                                            const uint32_t *__restrict__ tiles_touched,
                                                            uint32_t *__restrict__ tiles_sorted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < P) tiles_sorted[i] = tiles_touched[perm[i]];
}

// One lane per depth-ordered triangle.  Triangles covering up to SMALL tiles are emitted by their own lane; larger
// ones (stress scenes where a triangle spans thousands of tiles) are emitted cooperatively by the whole wave so that
// a single lane never serialises a long loop.  Tiles of one triangle are emitted row-major like the reference's
// loop (rasterizer.cu:63-73); the later sort is by tile id, so only the order BETWEEN triangles matters.
constexpr uint32_t SMALL = 32;

__global__ void __launch_bounds__(256) emit_instances_kernel(int P, int grid_x, GeometryStateView g, BinningStateView b)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool valid = i < P;
    const uint32_t tiles = valid ? g.tiles_sorted[i] : 0u;
    uint2 rect = {0u, 0u};
    uint32_t id = 0, off = 0;
    if (tiles > 0)
    {
        id = g.perm[i];
        rect = g.rect[id];
        off = g.offsets[i] - tiles; // exclusive prefix
    }
    const uint32_t minx = rect.x & 0xffffu, miny = rect.x >> 16, maxx = rect.y & 0xffffu, maxy = rect.y >> 16;
    if (tiles > 0 && tiles <= SMALL)
    {
        uint32_t o = off;
        for (uint32_t y = miny; y < maxy; y++)
            for (uint32_t x = minx; x < maxx; x++)
            {
                b.tile_unsorted[o] = y * grid_x + x;
                b.vals_unsorted[o] = id;
                o++;
            }
    }
    unsigned long long big = __ballot(tiles > SMALL);
    while (big)
    {
        const int j = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t t_minx = __shfl(minx, j), t_miny = __shfl(miny, j), t_maxx = __shfl(maxx, j);
        const uint32_t t_tiles = __shfl(tiles, j), t_off = __shfl(off, j), t_id = __shfl(id, j);
        const uint32_t w = t_maxx - t_minx;
        for (uint32_t k = lane; k < t_tiles; k += 64)
        {
            const uint32_t y = t_miny + k / w, x = t_minx + k % w;
            b.tile_unsorted[t_off + k] = y * grid_x + x;
            b.vals_unsorted[t_off + k] = t_id;
        }
    }
}

__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t N, const uint32_t *__restrict__ tile, uint2 *__restrict__ ranges)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t cur = tile[i];
    if (i == 0) ranges[cur].x = 0;
    else
    {
        const uint32_t prev = tile[i - 1];
        if (cur != prev)
        {
            ranges[prev].y = (uint32_t)i;
            ranges[cur].x = (uint32_t)i;
        }
    }
    if (i == N - 1) ranges[cur].y = (uint32_t)N;
}
} // namespace

// rocPRIM's default switches from merge sort to Onesweep radix sort above 1 Mi items; measured on MI355X (rocprofv3,
// profiles/r01_final_kernel_stats.csv) the merge path costs ten ~8 us merge passes at P = 1 M, about twice the four
// Onesweep passes, so the switch-over is lowered.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 128 * 1024>;

size_t ts_scan_temp_bytes(int32_t P)
{
    size_t scan = 0, sort = 0;
    if (P <= 0) return 0;
    (void)rocprim::inclusive_scan(nullptr, scan, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)P, rocprim::plus<uint32_t>());
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, sort, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)P, 0u, 32u);
    return scan > sort ? scan : sort;
}

size_t ts_sort_temp_bytes(int64_t N, int end_bit)
{
    size_t bytes = 0;
    if (N <= 0) return 0;
    (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)N, 0u, (unsigned)end_bit);
    return bytes;
}

// Step 1: (depth bits, id) -> perm.  Depth keys are view-space z of visible triangles (> 0, so the unsigned bit
// pattern is monotone) and 0 for culled ones, which emit nothing wherever they land.
hipError_t ts_sort_by_depth(const GeometryStateView &g, int32_t P, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    size_t bytes = g.scan_temp_bytes;
    return rocprim::radix_sort_pairs<SortConfig>(g.scan_temp, bytes, (const uint32_t *)g.depth, g.depth_sorted, g.ids, g.perm, (size_t)P,
                                     0u, 32u, s);
}

// Step 2: tiles_sorted = tiles_touched[perm], offsets = inclusive_scan(tiles_sorted).
hipError_t ts_scan_offsets(const GeometryStateView &g, int32_t P, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_tiles_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, g.perm, g.tiles_touched, g.tiles_sorted);
    size_t bytes = g.scan_temp_bytes;
    return rocprim::inclusive_scan(g.scan_temp, bytes, g.tiles_sorted, g.offsets, (size_t)P, rocprim::plus<uint32_t>(), s);
}

void ts_launch_emit_keys(int P, int grid_x, const GeometryStateView &g, const BinningStateView &b, hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(emit_instances_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, grid_x, g, b);
}

// Step 4: stable sort of the instances by tile id (end_bit = bits needed for the tile count).
hipError_t ts_sort_pairs(const BinningStateView &b, int64_t N, int end_bit, hipStream_t s)
{
    if (N <= 0) return hipSuccess;
    size_t bytes = b.sort_temp_bytes;
    return rocprim::radix_sort_pairs<SortConfig>(b.sort_temp, bytes, b.tile_unsorted, b.tile, b.vals_unsorted, b.vals, (size_t)N, 0u,
                                     (unsigned)end_bit, s);
}

void ts_launch_tile_ranges(int64_t N, const BinningStateView &b, const ImageStateView &im, hipStream_t s)
{
    if (N <= 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, N, b.tile, im.ranges);
}
