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

#include "ts2d_select.h"

namespace
{
__global__ void __launch_bounds__(SEL_BLOCK) sel_hist_kernel(const uint32_t *__restrict__ keys, size_t n, int pass, unsigned long long rank, SelState *st)
{
    uint32_t prefix;
    unsigned long long rem;
    sel_resolve(st, pass, rank, prefix, rem);
    const size_t base = (size_t)blockIdx.x * (SEL_BLOCK * SEL_ITEMS);
    sel_block_hist(st, pass, prefix, SEL_ITEMS, [&](int i, bool &valid) {
        const size_t k = base + (size_t)i * SEL_BLOCK + threadIdx.x;
        valid = k < n;
        return valid ? keys[k] : 0u;
    });
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
    // one pair of atomics per BLOCK: per wave, 4 x 157 blocks queued on two addresses took 21 us at 640 k keys
    __shared__ unsigned long long s_le[4];
    __shared__ uint32_t s_mn[4];
    if ((threadIdx.x & 63) == 0) { s_le[threadIdx.x >> 6] = le; s_mn[threadIdx.x >> 6] = mn; }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        le = s_le[0] + s_le[1] + s_le[2] + s_le[3];
        mn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
        if (le) atomicAdd(&st->count_le, le);
        if (mn != 0xffffffffu) atomicMax(&st->max_not_gt, ~mn);
    }
}

__global__ void __launch_bounds__(SEL_BLOCK) sel_threshold_kernel(size_t n, float q, const SelState *st, float *__restrict__ thr)
{
    const float v = sel_threshold_value(st, n, q);
    if (threadIdx.x == 0) *thr = v;
}
} // namespace

size_t ts_quantile_scratch_bytes() { return sizeof(SelState) + TS_ALIGN; }

// thr[0] = torch.quantile(keys as non-negative floats, q); `scratch` >= ts_quantile_scratch_bytes().  n >= 1.
void ts_quantile_threshold(const uint32_t *keys, size_t n, float q, void *scratch, float *thr, hipStream_t s)
{
    SelState *st = (SelState *)ts_align_up((size_t)scratch);
    ts_launch_zero_words((uint32_t *)st, sizeof(SelState) / 4, s);
    ts_quantile_passes(keys, n, q, scratch, 0, s);
    hipLaunchKernelGGL(sel_threshold_kernel, dim3(1), dim3(SEL_BLOCK), 0, s, n, q, st, thr);
}

// Passes first_pass .. 3 and the neighbour pass on a state whose earlier passes were filled by the caller's own kernels (ts2d_select.h: sel_block_hist);
// the caller's consumer kernel then takes the threshold with sel_threshold_value.  The state must have been zeroed before pass 0.
void ts_quantile_passes(const uint32_t *keys, size_t n, float q, void *scratch, int first_pass, hipStream_t s)
{
    SelState *st = (SelState *)ts_align_up((size_t)scratch);
    const unsigned long long lo = sel_rank_lo(n, q);
    const unsigned blocks = (unsigned)((n + SEL_BLOCK * SEL_ITEMS - 1) / (SEL_BLOCK * SEL_ITEMS));
    for (int pass = first_pass; pass < 4; pass++) hipLaunchKernelGGL(sel_hist_kernel, dim3(blocks), dim3(SEL_BLOCK), 0, s, keys, n, pass, lo, st);
    hipLaunchKernelGGL(sel_neighbour_kernel, dim3(blocks), dim3(SEL_BLOCK), 0, s, keys, n, lo, st);
}
size_t ts_quantile_state_words() { return sizeof(SelState) / 4; }
