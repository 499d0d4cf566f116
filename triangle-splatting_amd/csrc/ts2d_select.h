// ts2d_select.h -- the device side of the radix SELECT behind torch.quantile (select.hip): state layout, the per-block "which bin holds the rank"
// resolution, a block-level histogram step and the final threshold, shared with the kernels that produce the keys (depth_normal.hip fuses the first
// histogram pass into the kernel that writes them and the threshold into the kernel that consumes it: 13 -> 9 dependent launches on a 640 k-pixel image).
#pragma once
#include "ts2d_common.h"

namespace
{
constexpr int SEL_BLOCK = 256, SEL_ITEMS = 16; // keys per thread and launch
struct SelState
{
    uint32_t hist[4][256]; // digit totals of pass p (most significant first), among the keys that match passes 0 .. p - 1
    unsigned long long count_le; // keys <= the selected value
    uint32_t max_not_gt;         // ~(smallest key > the selected value), kept complemented so that the all-zero state means "none"
    uint32_t pad;
};

// (prefix, remaining rank) after `passes` passes, recomputed from the histograms by whoever needs it (256 threads, one block-wide scan per pass)
__device__ __forceinline__ void sel_resolve(const SelState *st, int passes, unsigned long long rank, uint32_t &prefix, unsigned long long &rem)
{
    __shared__ uint32_t s_pick;
    __shared__ unsigned long long s_before;
    prefix = 0u;
    rem = rank;
    __shared__ unsigned long long s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int p = 0; p < passes; p++)
    {
        // inclusive scan of the 256 digit totals over the block's 256 threads (wave scan + four wave totals); the bin that holds the remaining
        // rank is the one thread whose [exclusive, inclusive) interval contains it
        const unsigned long long cnt = st->hist[p][threadIdx.x];
        unsigned long long inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const unsigned long long up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        __syncthreads(); // (the previous pass's readers of s_pick / s_before / s_wave are done)
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        unsigned long long off = 0;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        inc += off;
        const unsigned long long exc = inc - cnt;
        if (cnt > 0 && exc <= rem && rem < inc) { s_pick = threadIdx.x; s_before = exc; }
        if (threadIdx.x == 255 && rem >= inc) { s_pick = 255u; s_before = exc; } // (a rank beyond the population: cannot happen for rank <= n - 1)
        __syncthreads();
        prefix |= s_pick << (24 - 8 * p);
        rem -= s_before;
    }
    __syncthreads();
}


// One block's contribution to pass `pass`'s histogram: every thread offers up to ITEMS keys through `key_at(slot, valid)`; LDS histogram, then at most
// 256 global adds.  For pass 0 nothing has to be resolved (prefix = 0, every key takes part).
template <typename KeyAt>
__device__ __forceinline__ void sel_block_hist(SelState *st, int pass, uint32_t prefix, int items, KeyAt key_at)
{
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (32 - 8 * pass));
    const int shift = 24 - 8 * pass;
    for (int i = 0; i < items; i++)
    {
        bool valid;
        const uint32_t v = key_at(i, valid);
        if (valid && (v & mask) == prefix) atomicAdd(&s_h[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&st->hist[pass][threadIdx.x], s_h[threadIdx.x]);
}

// rank floor(q (n - 1)) in float32, like torch.quantile's
__host__ __device__ __forceinline__ unsigned long long sel_rank_lo(size_t n, float q)
{
    const float rank = q * (float)(n - 1);
    unsigned long long lo = (unsigned long long)floorf(rank);
    return lo > n - 1 ? n - 1 : lo;
}

// torch.quantile(G, q), interpolation = "linear", from the finished state (all four passes + the neighbour pass): every thread of the block gets it
__device__ __forceinline__ float sel_threshold_value(const SelState *st, size_t n, float q)
{
    const unsigned long long lo = sel_rank_lo(n, q);
    uint32_t value;
    unsigned long long rem;
    sel_resolve(st, 4, lo, value, rem);
    const float rank = q * (float)(n - 1);
    unsigned long long hi = (unsigned long long)ceilf(rank);
    if (hi > n - 1) hi = n - 1;
    const float a = __uint_as_float(value);
    const float b = (hi == lo || st->count_le >= lo + 2) ? a : __uint_as_float(~st->max_not_gt); // rank lo + 1: the same value while it has duplicates, else the next larger key
    const float wgt = rank - (float)lo;
    return (wgt < 0.5f) ? a + wgt * (b - a) : b - (b - a) * (1.0f - wgt); // at::lerp
}
} // namespace
