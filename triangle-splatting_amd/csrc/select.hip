// select.hip -- torch.quantile(G, q) of n non-negative floats WITHOUT sorting them (ts_quantile_threshold, ts2d_common.h).
//
// The reference's DepthNormalLoss / SmoothnessLoss mask the steepest pixels with `G < torch.quantile(G, q)` (trainer_utils.py:196-199, 237-242): one
// number out of H x W values.  Rounds 3-5 obtained it from a full radix sort of the values (0.15 ms of the 0.23 ms the depth / normal loss took at
// 1080p).  A quantile needs two ORDER STATISTICS -- the values of rank floor(r) and ceil(r), r = q (n - 1) -- and the bit patterns of non-negative
// floats order like the values, so a most-significant-digit radix SELECT finds the first exactly: four passes, each a histogram of one 8-bit
// digit over the keys that match the digits chosen so far (per-block LDS histogram, 256 global adds per block), the bin that holds the rank picked by
// EVERY block of the next launch for itself from the 256 totals (no single-block step, no host round trip).  One more pass counts the keys <= that
// value and takes the minimum of the larger ones: the value of rank + 1 is the same value when it has duplicates, that minimum otherwise.
// Exact by construction (integers only until the final lerp, which is at::lerp's expression as before).  Five streaming launches over 4 n bytes.
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

__global__ void __launch_bounds__(SEL_BLOCK) sel_hist_kernel(const uint32_t *__restrict__ keys, size_t n, int pass, unsigned long long rank, SelState *st)
{
    __shared__ uint32_t s_h[256];
    uint32_t prefix;
    unsigned long long rem;
    sel_resolve(st, pass, rank, prefix, rem);
    s_h[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t mask = pass == 0 ? 0u : (0xffffffffu << (32 - 8 * pass));
    const int shift = 24 - 8 * pass;
    const size_t base = (size_t)blockIdx.x * (SEL_BLOCK * SEL_ITEMS);
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; i++)
    {
        const size_t k = base + (size_t)i * SEL_BLOCK + threadIdx.x;
        if (k < n)
        {
            const uint32_t v = keys[k];
            if ((v & mask) == prefix) atomicAdd(&s_h[(v >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (s_h[threadIdx.x]) atomicAdd(&st->hist[pass][threadIdx.x], s_h[threadIdx.x]);
}

__global__ void __launch_bounds__(SEL_BLOCK) sel_neighbour_kernel(const uint32_t *__restrict__ keys, size_t n, unsigned long long rank, SelState *st)
{
    uint32_t value;
    unsigned long long rem;
    sel_resolve(st, 4, rank, value, rem);
    unsigned long long le = 0;
    uint32_t mn = 0xffffffffu;
    const size_t base = (size_t)blockIdx.x * (SEL_BLOCK * SEL_ITEMS);
#pragma unroll
    for (int i = 0; i < SEL_ITEMS; i++)
    {
        const size_t k = base + (size_t)i * SEL_BLOCK + threadIdx.x;
        if (k < n)
        {
            const uint32_t v = keys[k];
            le += v <= value ? 1u : 0u;
            mn = v > value ? min(mn, v) : mn;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        le += __shfl_xor(le, o);
        mn = min(mn, (uint32_t)__shfl_xor((int)mn, o));
    }
    if ((threadIdx.x & 63) == 0)
    {
        if (le) atomicAdd(&st->count_le, le);
        if (mn != 0xffffffffu) atomicMax(&st->max_not_gt, ~mn);
    }
}

// torch.quantile(G, q), interpolation = "linear": rank = q (n - 1) in float32 like torch, lerp between the two neighbours (at::lerp)
__global__ void __launch_bounds__(SEL_BLOCK) sel_threshold_kernel(size_t n, float q, unsigned long long rank_lo, const SelState *st, float *__restrict__ thr)
{
    uint32_t value;
    unsigned long long rem;
    sel_resolve(st, 4, rank_lo, value, rem);
    if (threadIdx.x != 0) return;
    const float rank = q * (float)(n - 1);
    const unsigned long long lo = (unsigned long long)floorf(rank);
    unsigned long long hi = (unsigned long long)ceilf(rank);
    if (hi > n - 1) hi = n - 1;
    const float a = __uint_as_float(value);
    // rank lo + 1: the same value while it has duplicates beyond rank lo, else the next larger key
    const float b = (hi == lo || st->count_le >= lo + 2) ? a : __uint_as_float(~st->max_not_gt);
    const float wgt = rank - (float)lo;
    *thr = (wgt < 0.5f) ? a + wgt * (b - a) : b - (b - a) * (1.0f - wgt);
}
} // namespace

size_t ts_quantile_scratch_bytes() { return sizeof(SelState) + TS_ALIGN; }

// thr[0] = torch.quantile(keys as non-negative floats, q); `scratch` >= ts_quantile_scratch_bytes().  n >= 1.
void ts_quantile_threshold(const uint32_t *keys, size_t n, float q, void *scratch, float *thr, hipStream_t s)
{
    SelState *st = (SelState *)ts_align_up((size_t)scratch);
    const float rank = q * (float)(n - 1); // float32, like torch.quantile's rank
    unsigned long long lo = (unsigned long long)floorf(rank);
    if (lo > n - 1) lo = n - 1;
    ts_launch_zero_words((uint32_t *)st, sizeof(SelState) / 4, s);
    const unsigned blocks = (unsigned)((n + SEL_BLOCK * SEL_ITEMS - 1) / (SEL_BLOCK * SEL_ITEMS));
    for (int pass = 0; pass < 4; pass++) hipLaunchKernelGGL(sel_hist_kernel, dim3(blocks), dim3(SEL_BLOCK), 0, s, keys, n, pass, lo, st);
    hipLaunchKernelGGL(sel_neighbour_kernel, dim3(blocks), dim3(SEL_BLOCK), 0, s, keys, n, lo, st);
    hipLaunchKernelGGL(sel_threshold_kernel, dim3(1), dim3(SEL_BLOCK), 0, s, n, q, lo, st, thr);
}
