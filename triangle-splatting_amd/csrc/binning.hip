// binning.hip -- depth ordering, instance emission, tile grouping and tile ranges, all hand-written for gfx950.
//
// Result contract (integer-exact, what the blend kernels and the parity tests rely on): the instance list is ordered
// by (tile id, depth bit pattern, triangle id) -- exactly what the reference obtains with cub::DeviceScan::InclusiveSum
// + duplicateWithKeys + one stable cub::DeviceRadixSort::SortPairs over N 64-bit (tile << 32 | depth) keys +
// identifyTileRanges (R2D/src/rasterizer.cu:186, 37-75, 210-218, 79-99).
//
// How it is obtained here (same order, ~4.5x less sort traffic; N ~ 4.6 x P for the headline scene):
//   1. stable radix sort of the P triangles by their 32-bit depth key (values = ascending ids): 3-4 passes x P pairs; the first histogram
//      also produces N = sum(tiles_touched) (the one value the host reads back) and finds out whether the fourth pass can be skipped;
//   2. tiles_touched gathered in that order + block sums;
//   3. per block: wave64 prefix scan (DPP) of the tile counts on top of the block sums in front -> instance slots, and (tile, id)
//      instances emitted in depth order through LDS; the same kernel clears the tile ranges and the contribution statistics;
//   4. stable radix sort of the N instances by TILE ID ONLY (13 bits at 1080p -> 2 passes x N x 8 B instead of
//      6 passes x N x 12 B); stability keeps the depth order (and the id order among equal depths) inside a tile;
//   5. tile ranges from the sorted tile ids.
//
// One radix pass (digit of up to 8 bits) = two kernels, no look-back spinning.  Two flavours of the first one:
//   rs_hist_direct  (sorts of up to 48 slabs = 12.6 M pairs: everything the headline runs) one workgroup per chunk of 2048 / 4096 pairs
//               counts its digits in a 1 KB LDS table (ds_add_u32), stores the 256 counts as a raw table row and adds them to its slab's
//               totals with fire-and-forget atomics.  Nobody waits for anybody: the scatter kernel works out its prefixes itself
//               (<= 63 rows of its slab + the slabs' totals, requested while its keys are on their way);
//   rs_hist     (larger sorts) the workgroup that arrives LAST in its slab of 64 chunks (one atomic ticket; the counts travel as
//               write-through stores and L1-bypassing loads, so no L2 write-back fence is needed) turns the slab's rows into
//               exclusive column prefixes, and the last slab to finish does the same over the slab totals and over the 256 digit totals.
//               Its cost does not grow with the slab count, but the elected block walks seven dependent memory round trips alone:
//               18 us at 1 M keys, which is why the small sorts left it;
//   rs_scatter  the workgroup re-reads its chunk (each wave a contiguous quarter, 64 pairs per step): the lanes holding equal
//               digits find each other with one ballot per digit bit (wave64 match), rank = v_mbcnt of the match mask on top of the
//               digit's running count; the pairs are parked in LDS in chunk-local sorted order and leave as coalesced runs.  Ranks
//               follow lane order, steps follow list order, waves follow chunk order: stable by construction.
// The same rule removed the elected blocks from the scan (raw block sums, added up by the emission blocks) and from the census (published
// by block 0 of the first scatter).  The round-1 rocPRIM calls (radix_sort_pairs, inclusive_scan) survive only as the comparators of
// tests/test_binning_gpu.py.
#include "ts2d_common.h"
#include "ts2d_wave.h"
#include "ts2d_support.h"

namespace
{
constexpr int NB = TS_RS_BINS;

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane) // DPP row shifts + two row broadcasts
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true); // row_bcast:15 -> rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true); // row_bcast:31 -> rows 2 and 3
    return (uint32_t)x;
}

// Elects the block that arrives last at `ticket` among `count` arrivals; the elected block resets the ticket for the next
// launch and returns true on all of its threads.  Data handed to the elected block travels as write-through (sc0 sc1) stores and
// L1-bypassing (sc1) loads on both sides (peer_store / peer_load): with every store drained (s_waitcnt vmcnt(0)) before the ticket
// is taken, no L2 write-back fence is needed (MI355X_MICROARCH.md, "valid forms": a release fence per block costs 2-6 us).
// This is a HARDWARE contract of gfx950 (write-through sc0 sc1 stores + drain on the producer, sc1 loads on the consumer, both relaxed in
// the language's memory model), not something the HIP memory model promises: hence the target check below, and
// tests/test_binning_gpu.py::test_last_arrival_handoffs_under_uneven_load hammers every hand-off next to a noisy neighbour stream.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "last_arrival() relies on gfx950's write-through store / L1-bypassing load behaviour; re-validate before building for another target"
#endif
__device__ __forceinline__ bool last_arrival(uint32_t *ticket, uint32_t count)
{
    __shared__ bool elected;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        elected = (t == count - 1u);
        if (elected) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return elected;
}
template <typename T>
__device__ __forceinline__ T peer_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void peer_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// triangle ids in (depth, id) order: the 4th pass's output, or the 3rd's when the 4th was skipped (written by an earlier launch)
__device__ __forceinline__ const uint32_t *sorted_ids(const GeometryStateView &g) { return *g.top_const ? g.sv[0] : g.sv[1]; }

// Sync-free forward (ts2d_forward): the pair count lives on the device.  `n_dev` (null on the synchronous path) points at the
// 64-bit instance count the scan left behind; a count above `n` (the capacity the buffers were carved for) renders nothing and is
// reported through the state's status word.  The launch covers the capacity; blocks past the actual count return at once.
template <int CH>
__device__ __forceinline__ bool resolve_count(const unsigned long long *n_dev, int64_t &n, RadixScratchView &r)
{
    if (n_dev)
    {
        const unsigned long long live = *n_dev;
        n = (live <= (unsigned long long)n) ? (int64_t)live : 0;
        r.chunks = (int)((n + CH - 1) / CH);
        r.slabs = (r.chunks + 63) / 64;
    }
    return (int)blockIdx.x < 8 * ((r.chunks + 7) / 8);
}
// Which chunk a workgroup of a radix pass works on.  Workgroups go to the 8 XCDs round-robin (block b -> XCD b % 8); XCD x takes the chunks
// [x n / 8, (x + 1) n / 8): consecutive chunks write adjacent pieces of every digit's run (32 - 64 elements = one or two cache lines each),
// and with ONE XCD behind both halves of a shared line the two partial writes meet in one L2 instead of two (round 4: tile sort 73 -> 69 us,
// depth sort 48 -> 45 us against chunk = blockIdx, -DTS_RS_XCD_ROUND_ROBIN).  The partition follows the LIVE chunk count, so a launch that
// covers a larger capacity (speculative / sync-free forward) stays balanced over the XCDs.
__device__ __forceinline__ int rs_chunk_of_block(int nchunks)
{
#ifdef TS_RS_XCD_ROUND_ROBIN
    return (int)blockIdx.x < nchunks ? (int)blockIdx.x : -1;
#else
    const int q = nchunks >> 3, r = nchunks & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    return i < q + (x < r ? 1 : 0) ? x * q + min(x, r) + i : -1;
#endif
}

// Digit counts of every chunk, and -- by the blocks that arrive last -- their prefixes: the last block of a slab (64 chunks)
// turns the slab's rows into exclusive column prefixes and its totals; the last slab to finish turns the slab totals into
// their prefix over the slabs and forms the exclusive prefix of the 256 digit totals.  One launch, no spinning.
//
// The FIRST pass of the depth sort (CENSUS) also takes stock of what it reads anyway (round 3: two launches and one pass fewer per step):
//   * N = sum(tiles_touched), the instance count the host is waiting for (the reference hands num_rendered to the host,
//     rasterizer.cu:189-191); it does not depend on the depth order, so it leaves as soon as this kernel is done -- through a pinned host
//     word -- while the rest of the sort runs (round 2 spent a kernel of its own on it);
//   * which key bits differ between VISIBLE triangles (culled ones carry key 0 and emit nothing wherever they land): depths are positive
//     floats, and when they all share their top byte -- sign + seven exponent bits: every scene whose depths span less than a factor
//     of four -- the fourth pass has nothing to order.  The verdict goes to `census->top_const`; the last pass's kernels return at once
//     when it is set and the consumers of the order take the third pass's output (sorted_ids()).
struct DepthCensus
{
    const uint32_t *tiles_touched; // per key
    unsigned long long *chunk_sum; // per chunk (scratch)
    uint32_t *chunk_or, *chunk_and; // per chunk (scratch)
    unsigned long long *n_out;     // device: where the scan will leave N as well
    unsigned long long *host_out;  // pinned host word or null
    uint32_t *top_const;           // device flag
    uint32_t force_varying;        // lab library only (ts2d_lab_force_depth_pass4): key bits reported as varying whatever the scene holds
};
__device__ __forceinline__ bool pass_skipped(const uint32_t *skip_flag) { return skip_flag && peer_load(skip_flag) != 0u; }

template <bool CENSUS, int CH>
__global__ void __launch_bounds__(256) rs_hist_kernel(const uint32_t *__restrict__ keys, int64_t n, const unsigned long long *n_dev, int shift,
                                                       uint32_t mask, RadixScratchView r, DepthCensus census, const uint32_t *skip_flag)
{
    __shared__ uint32_t bins[NB];
    __shared__ unsigned long long csum[4];
    __shared__ uint32_t cor[4], cand[4];
    if (!resolve_count<CH>(n_dev, n, r)) return;
    if (!CENSUS && pass_skipped(skip_flag)) return;
    const int t = threadIdx.x, chunk = rs_chunk_of_block(r.chunks);
    if (chunk < 0) return;
    bins[t] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)chunk * CH;
    unsigned long long tsum = 0;
    uint32_t kor = 0u, kand = 0xFFFFFFFFu;
#pragma unroll 4
    for (int b = 0; b < CH / 256; b++)
    {
        const int64_t i = base + 256 * b + t;
        if (i < n)
        {
            const uint32_t k = keys[i];
            atomicAdd(&bins[(k >> shift) & mask], 1u);
            if (CENSUS)
            {
                tsum += census.tiles_touched[i];
                if (k != 0u) { kor |= k; kand &= k; }
            }
        }
    }
    if (CENSUS)
    {
        for (int o = 32; o > 0; o >>= 1)
        {
            tsum += __shfl_xor(tsum, o);
            kor |= __shfl_xor(kor, o);
            kand &= __shfl_xor(kand, o);
        }
        if ((t & 63) == 0) { csum[t >> 6] = tsum; cor[t >> 6] = kor; cand[t >> 6] = kand; }
    }
    __syncthreads();
    peer_store(r.table + (size_t)chunk * NB + t, bins[t]);
    if (CENSUS && t == 0)
    {
        peer_store(census.chunk_sum + chunk, csum[0] + csum[1] + csum[2] + csum[3]);
        peer_store(census.chunk_or + chunk, cor[0] | cor[1] | cor[2] | cor[3]);
        peer_store(census.chunk_and + chunk, cand[0] & cand[1] & cand[2] & cand[3]);
    }

    const int slab = chunk >> 6, c0 = slab * 64, c1 = min(r.chunks, c0 + 64);
    if (!last_arrival(r.tickets + 2 + slab, (uint32_t)(c1 - c0))) return;
    uint32_t run = 0;
    {
        // the elected block is alone on the launch's critical path: all 64 rows of the slab are requested before the first one is used
        // (one memory round trip instead of four)
        uint32_t v[64];
#pragma unroll
        for (int k = 0; k < 64; k++) v[k] = (c0 + k < c1) ? peer_load(r.table + (size_t)(c0 + k) * NB + t) : 0u;
#pragma unroll
        for (int k = 0; k < 64; k++)
        {
            if (c0 + k < c1) r.table[(size_t)(c0 + k) * NB + t] = run;
            run += v[k];
        }
    }
    peer_store(r.slabtot + (size_t)slab * NB + t, run);

    if (!last_arrival(r.tickets, (uint32_t)r.slabs)) return;
    uint32_t total = 0;
    for (int s = 0; s < r.slabs; s += 32)
    {
        uint32_t v[32];
#pragma unroll
        for (int k = 0; k < 32; k++) v[k] = (s + k < r.slabs) ? peer_load(r.slabtot + (size_t)(s + k) * NB + t) : 0u;
#pragma unroll
        for (int k = 0; k < 32; k++)
        {
            if (s + k < r.slabs) r.slabtot[(size_t)(s + k) * NB + t] = total;
            total += v[k];
        }
    }
    {
        // exclusive prefix of the 256 digit totals: wave64 DPP scan + the preceding waves' totals
        __shared__ uint32_t wtot[4];
        const uint32_t inc = wave_inclusive_scan(total, t & 63);
        if ((t & 63) == 63) wtot[t >> 6] = inc;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < (t >> 6); w++) before += wtot[w];
        r.binbase[t] = before + inc - total;
    }
    if (CENSUS)
    {
        // this block arrived last of all: every chunk's census is visible (same hand-off as the digit counts)
        unsigned long long sum = 0;
        uint32_t o = 0u, a = 0xFFFFFFFFu;
        for (int c = t; c < r.chunks; c += 256)
        {
            sum += peer_load(census.chunk_sum + c);
            o |= peer_load(census.chunk_or + c);
            a &= peer_load(census.chunk_and + c);
        }
        for (int d = 32; d > 0; d >>= 1)
        {
            sum += __shfl_xor(sum, d);
            o |= __shfl_xor(o, d);
            a &= __shfl_xor(a, d);
        }
        __syncthreads();
        if ((t & 63) == 0) { csum[t >> 6] = sum; cor[t >> 6] = o; cand[t >> 6] = a; }
        __syncthreads();
        if (t == 0)
        {
            const unsigned long long N = csum[0] + csum[1] + csum[2] + csum[3];
            const uint32_t varying = ((cor[0] | cor[1] | cor[2] | cor[3]) ^ (cand[0] & cand[1] & cand[2] & cand[3])) | census.force_varying;
            *census.n_out = N;
            peer_store(census.top_const, (varying >> 24) == 0u ? 1u : 0u); // no visible triangle at all: or = 0, and = ~0 -> varying = ~0 -> not set
            // pinned, device-visible host word: the host reads it after the event recorded behind this kernel (no copy kernel in between)
            if (census.host_out) __hip_atomic_store(census.host_out, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Ticket-free histogram (round 3): a pass of the hierarchical version above is a chain of seven dependent memory round trips (keys ->
// counts -> ticket -> slab rows -> ticket -> slab totals -> prefixes) that one elected block walks alone while the chip idles; at 1 M keys
// that chain IS the kernel (18 us for 4 MB of keys).  Here a block leaves its 256 counts as a table row and adds them to its slab's totals
// with fire-and-forget atomics, and that is all; the scatter kernel that follows works out the three prefixes it needs from the rows
// (<= 63 rows of its slab + the slabs' totals, loaded while its keys are on their way).
constexpr int TS_DIRECT_MAX_SLABS = 48; // every scatter block reads the totals of all slabs: beyond this the hierarchical pass is cheaper
template <int CH>
__global__ void __launch_bounds__(256) rs_hist_direct_kernel(const uint32_t *__restrict__ keys, int64_t n, const unsigned long long *n_dev, int shift,
                                                              uint32_t mask, RadixScratchView r, uint32_t *__restrict__ acc, const uint32_t *skip_flag)
{
    __shared__ uint32_t bins[NB];
    if (!resolve_count<CH>(n_dev, n, r)) return;
    if (pass_skipped(skip_flag)) return;
    const int t = threadIdx.x, chunk = rs_chunk_of_block(r.chunks);
    if (chunk < 0) return;
    bins[t] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)chunk * CH;
#pragma unroll
    for (int b = 0; b < CH / 1024; b++) // one dwordx4 per thread and round (the counts do not care about the order inside the chunk)
    {
        const int64_t i = base + 1024 * b + 4 * t;
        if (i + 3 < n)
        {
            const uint4 q = *(const uint4 *)(keys + i);
            atomicAdd(&bins[(q.x >> shift) & mask], 1u);
            atomicAdd(&bins[(q.y >> shift) & mask], 1u);
            atomicAdd(&bins[(q.z >> shift) & mask], 1u);
            atomicAdd(&bins[(q.w >> shift) & mask], 1u);
        }
        else
            for (int k = 0; k < 4; k++)
                if (i + k < n) atomicAdd(&bins[(keys[i + k] >> shift) & mask], 1u);
    }
    __syncthreads();
    const uint32_t c = bins[t];
    r.table[(size_t)chunk * NB + t] = c;
    if (c) __hip_atomic_fetch_add(acc + (size_t)(chunk >> 6) * NB + t, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The depth sort's first histogram, ticket-free like the one above, with the census riding along: a block leaves its part of N and of the
// key-bit OR / OR-of-complements as three per-chunk words; block 0 of the scatter kernel that follows adds them up and publishes N (device
// word + pinned host word) and the top-byte verdict (`top_const`).  No ticket at all: electing a publisher HERE costs one same-address
// atomic per block, and 488 of those in a row take longer (17 us) than the histogram itself.  Large scenes (more than
// TS_DIRECT_MAX_SLABS slabs) keep rs_hist_kernel<true>.
template <int CH>
__global__ void __launch_bounds__(256) rs_hist_census_direct_kernel(const uint32_t *__restrict__ keys, int64_t n, uint32_t mask, RadixScratchView r,
                                                                     uint32_t *__restrict__ acc, DepthCensus census)
{
    __shared__ uint32_t bins[NB];
    __shared__ unsigned long long csum[4];
    __shared__ uint32_t cor[4], cnand[4];
    const int t = threadIdx.x, chunk = rs_chunk_of_block(r.chunks);
    if (chunk < 0) return;
    bins[t] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)chunk * CH;
    unsigned long long tsum = 0;
    uint32_t kor = 0u, knand = 0u;
#pragma unroll
    for (int b = 0; b < CH / 256; b++)
    {
        const int64_t i = base + 256 * b + t;
        if (i < n)
        {
            const uint32_t k = keys[i];
            atomicAdd(&bins[k & mask], 1u);
            tsum += census.tiles_touched[i];
            if (k != 0u) { kor |= k; knand |= ~k; }
        }
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        tsum += __shfl_xor(tsum, o);
        kor |= __shfl_xor(kor, o);
        knand |= __shfl_xor(knand, o);
    }
    if ((t & 63) == 0) { csum[t >> 6] = tsum; cor[t >> 6] = kor; cnand[t >> 6] = knand; }
    __syncthreads();
    const uint32_t c = bins[t];
    r.table[(size_t)chunk * NB + t] = c;
    if (c) __hip_atomic_fetch_add(acc + (size_t)(chunk >> 6) * NB + t, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0)
    {
        census.chunk_sum[chunk] = csum[0] + csum[1] + csum[2] + csum[3];
        census.chunk_or[chunk] = cor[0] | cor[1] | cor[2] | cor[3];
        census.chunk_and[chunk] = ~(cnand[0] | cnand[1] | cnand[2] | cnand[3]);
    }
}

// Block 0 of the first scatter: the chunks' census words -> N, top_const, the pinned host word (see above).
__device__ __forceinline__ void publish_census(const DepthCensus &census, int chunks, int t)
{
    __shared__ unsigned long long psum[4];
    __shared__ uint32_t por[4], pand[4];
    unsigned long long sum = 0;
    uint32_t o = 0u, a = 0xFFFFFFFFu;
    for (int c = t; c < chunks; c += 256)
    {
        sum += census.chunk_sum[c];
        o |= census.chunk_or[c];
        a &= census.chunk_and[c];
    }
    for (int d = 32; d > 0; d >>= 1)
    {
        sum += __shfl_xor(sum, d);
        o |= __shfl_xor(o, d);
        a &= __shfl_xor(a, d);
    }
    if ((t & 63) == 0) { psum[t >> 6] = sum; por[t >> 6] = o; pand[t >> 6] = a; }
    __syncthreads();
    if (t == 0)
    {
        const unsigned long long N = psum[0] + psum[1] + psum[2] + psum[3];
        const uint32_t varying = ((por[0] | por[1] | por[2] | por[3]) ^ (pand[0] & pand[1] & pand[2] & pand[3])) | census.force_varying;
        *census.n_out = N;
        *census.top_const = (varying >> 24) == 0u ? 1u : 0u; // no visible triangle at all: or = 0, and = ~0 -> varying = ~0 -> not set
        // pinned, device-visible host word: the host reads it after the event recorded behind this kernel (no copy kernel in between)
        if (census.host_out) __hip_atomic_store(census.host_out, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One workgroup = one chunk of CH pairs; wave w owns the w-th quarter (KB steps of 64 consecutive pairs, held in registers).
//   1. wave-local stable ranks: per step the lanes holding equal digits find each other with `nbits` ballots (wave64 match),
//      rank = v_mbcnt of the match mask on top of the digit's running count in the wave's LDS counters;
//   2. thread d turns the four waves' counts of digit d into the chunk-local start of every (wave, digit) run (wave64 DPP scan
//      + the three other waves' totals) and the distance from the chunk-local order to the digit's global run;
//   3. every pair is parked in LDS at its chunk-local sorted position, and the chunk leaves in that order: consecutive threads
//      write consecutive addresses inside each digit's run (scattering straight from registers costs a 32-64 B fabric write
//      per 4-byte store on this chip: measured 2x slower than rocPRIM; staged, the stores are coalesced runs).
// DIRECT: the pass's histogram was rs_hist_direct_kernel; `acc` holds the slabs' digit totals and the table rows are raw counts.
// `acc_clear` (either flavour): the other totals buffer, cleared here for the next pass's histogram (nobody reads it any more).
// TWO_PHASE (the 4096-pair chunks of the instance sort): keys and values pass through ONE staging array one after the other, and the values
// are only loaded once the keys have been ranked.  Half the LDS and 16 registers less at the peak put seven workgroups on a CU instead of
// four: the 1126 chunks of the headline's instance list are then resident in one round (4 x 256 slots had left 102 of them for a second).
template <bool IDENTITY_VALUES, int CH, bool DIRECT, bool CENSUS, bool TWO_PHASE>
__device__ __forceinline__ void rs_scatter_body(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin, uint32_t *__restrict__ kout,
                                                uint32_t *__restrict__ vout, int64_t n, const unsigned long long *n_dev, int shift, int nbits,
                                                RadixScratchView r, const uint32_t *skip_flag, const uint32_t *__restrict__ acc,
                                                uint32_t *__restrict__ acc_clear, const DepthCensus &census)
{
    constexpr int KB = CH / 256; // steps per wave
    if (!resolve_count<CH>(n_dev, n, r)) return;
    if (pass_skipped(skip_flag)) return;
    if (acc_clear && (int)blockIdx.x < r.slabs) acc_clear[(size_t)blockIdx.x * NB + threadIdx.x] = 0u;
    constexpr int SK = (!TWO_PHASE && CH < 8 * NB) ? 8 * NB : CH; // the DIRECT prefix exchange borrows 8 (TWO_PHASE: 12) x 256 words of stage_k
    constexpr int SV = TWO_PHASE ? 4 : (CH < 4 * NB ? 4 * NB : CH); // ... and 4 x 256 words of stage_v (TWO_PHASE: all 12 x 256 of stage_k)
    __shared__ __attribute__((aligned(16))) uint32_t stage_k[SK], stage_v[SV];
    static_assert(!TWO_PHASE || CH >= 12 * NB, "TWO_PHASE borrows 12 x 256 words of stage_k");
    static_assert(!TWO_PHASE || (!IDENTITY_VALUES && !CENSUS), "TWO_PHASE is the instance sort's flavour");
    uint32_t *const xtotal = TWO_PHASE ? stage_k + 8 * NB : stage_v; // where the four waves' partial digit totals meet
    __shared__ uint32_t wcnt[4][NB]; // per-wave digit counts, then the chunk-local start of the (wave, digit) run
    __shared__ int32_t gdelta[NB];   // global run start of the digit minus its chunk-local start
    __shared__ uint32_t wtot[4], gtot[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int chunk = rs_chunk_of_block(r.chunks);
    if (chunk < 0) return;
    const uint32_t mask = (1u << nbits) - 1u;
    const int64_t base = (int64_t)chunk * CH + (int64_t)wave * (CH / 4);
    const int64_t here = n - base;                                        // pairs from this wave's first one to the end of the array
    const int mine = here >= CH / 4 ? CH / 4 : (here > 0 ? (int)here : 0); // ... of which this wave holds the first `mine`
    const uint32_t *kin_w = kin + base, *vin_w = IDENTITY_VALUES ? nullptr : vin + base;
    uint32_t key[KB], val[KB], rk[KB];
#pragma unroll
    for (int b = 0; b < KB; b++)
    {
        const int i = 64 * b + lane;
        key[b] = 0xFFFFFFFFu;
        val[b] = 0u;
        if (i < mine)
        {
            key[b] = kin_w[i];
            if (!TWO_PHASE) val[b] = IDENTITY_VALUES ? (uint32_t)(base + i) : vin_w[i];
        }
    }
    // DIRECT: how many pairs of each digit sit in earlier chunks of this slab, in earlier slabs, and in all slabs.  Wave w takes every
    // fourth row, lane l the digits 4l .. 4l + 3 (one dwordx4 per row: at most 16 + 12 loads per lane, all requested here, behind the keys);
    // the four waves' partial sums meet in LDS after the ranking (the staging arrays are still free then).
    uint4 p_within = make_uint4(0u, 0u, 0u, 0u), p_before = p_within, p_total = p_within;
    if (DIRECT)
    {
        const int slab = chunk >> 6, c0 = slab * 64;
        const uint4 *tab4 = (const uint4 *)r.table + (size_t)c0 * (NB / 4) + lane;
        const uint4 *acc4 = (const uint4 *)acc + lane;
#pragma unroll 4
        for (int k = 0; k < 16; k++)
        {
            const int c = wave + 4 * k;
            if (c0 + c < chunk)
            {
                const uint4 v = tab4[(size_t)c * (NB / 4)];
                p_within.x += v.x; p_within.y += v.y; p_within.z += v.z; p_within.w += v.w;
            }
        }
#pragma unroll 4
        for (int k = 0; k < TS_DIRECT_MAX_SLABS / 4; k++)
        {
            const int sl = wave + 4 * k;
            if (sl < r.slabs)
            {
                const uint4 v = acc4[(size_t)sl * (NB / 4)];
                p_total.x += v.x; p_total.y += v.y; p_total.z += v.z; p_total.w += v.w;
                if (sl < slab) { p_before.x += v.x; p_before.y += v.y; p_before.z += v.z; p_before.w += v.w; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB / 64; k++) wcnt[wave][lane + 64 * k] = 0u;
    wave_lds_order();
    uint32_t *cnt = wcnt[wave];
#pragma unroll
    for (int b = 0; b < KB; b++)
    {
        const bool valid = 64 * b + lane < mine;
        const uint32_t d = (key[b] >> shift) & mask;
        // lanes whose digit differs from mine in some bit: (ballot of bit i) xor (my bit i, sign-extended), or-ed over the bits, in two 32-bit
        // halves -- three instructions per bit and half (round 5; the select form `m &= one ? bb : ~bb` compiled to ~100 instructions per step,
        // and a launch of a few resident workgroups per SIMD spends a good part of its time issuing exactly these)
        const unsigned long long vm = ballot64(valid);
        uint32_t mis_lo = ~(uint32_t)vm, mis_hi = ~(uint32_t)(vm >> 32);
        // all eight bits, unrolled, whatever `nbits` is (the digit's bits above it are zero in every lane and cost a ballot that changes nothing): with
        // the bit index a compile-time constant a bit is four instructions (v_bfe_i32, the compare behind the ballot, two fused xor-or); as a loop
        // over the run-time `nbits` it was twelve (shift by an SGPR, select, loop control, two s_nop) -- 96 of the ~135 instructions of a 64-pair
        // step, in kernels whose time IS this ranking (round 6: profiles/r06_rank_unroll.txt)
#pragma unroll
        for (int bit = 0; bit < 8; bit++)
        {
            const unsigned long long bb = ballot64((d >> bit) & 1u);
            const uint32_t e = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit, 1); // all ones when my bit is set
            mis_lo |= (uint32_t)bb ^ e;
            mis_hi |= (uint32_t)(bb >> 32) ^ e;
        }
        const uint32_t m_lo = ~mis_lo, m_hi = ~mis_hi; // the valid lanes that hold my digit
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const uint32_t c = (uint32_t)(__popc(m_lo) + __popc(m_hi));
        uint32_t seen = 0;
        if (valid) seen = cnt[d];
        wave_lds_order(); // every lane has read its digit's count before the group leaders advance it
        if (valid && rank == c - 1u) cnt[d] = seen + c;
        wave_lds_order();
        rk[b] = seen + rank;
    }
    if (DIRECT)
    {
        *(uint4 *)(stage_k + wave * NB + 4 * lane) = p_within;
        *(uint4 *)(stage_k + (4 + wave) * NB + 4 * lane) = p_before;
        *(uint4 *)(xtotal + wave * NB + 4 * lane) = p_total;
    }
    __syncthreads();
    {
        // thread t = digit t
        uint32_t d_within = 0u, d_before = 0u, d_total = 0u;
        if (DIRECT)
        {
#pragma unroll
            for (int w = 0; w < 4; w++)
            {
                d_within += stage_k[w * NB + t];
                d_before += stage_k[(4 + w) * NB + t];
                d_total += xtotal[w * NB + t];
            }
        }
        const uint32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t], c3 = wcnt[3][t];
        const uint32_t tot = c0 + c1 + c2 + c3;
        const uint32_t inc = wave_inclusive_scan(tot, lane);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t dbase = inc - tot;
        for (int w = 0; w < wave; w++) dbase += wtot[w];
        wcnt[0][t] = dbase; wcnt[1][t] = dbase + c0; wcnt[2][t] = dbase + c0 + c1; wcnt[3][t] = dbase + c0 + c1 + c2;
        uint32_t g;
        if (DIRECT)
        {
            // exclusive prefix of the digit totals over the digits, the same way as the chunk-local one above
            const uint32_t ginc = wave_inclusive_scan(d_total, lane);
            if (lane == 63) gtot[wave] = ginc;
            __syncthreads();
            uint32_t gbase = ginc - d_total;
            for (int w = 0; w < wave; w++) gbase += gtot[w];
            g = gbase + d_before + d_within;
        }
        else g = r.binbase[t] + r.slabtot[(size_t)(chunk >> 6) * NB + t] + r.table[(size_t)chunk * NB + t];
        gdelta[t] = (int32_t)(g - dbase);
    }
    __syncthreads();
    const int64_t left = n - (int64_t)chunk * CH;
    const int count = left < CH ? (int)left : CH;
    if (TWO_PHASE)
    {
#pragma unroll
        for (int b = 0; b < KB; b++)
            if (64 * b + lane < mine)
            {
                rk[b] += wcnt[wave][(key[b] >> shift) & mask]; // chunk-local sorted position, kept for the values
                stage_k[rk[b]] = key[b];
            }
#pragma unroll
        for (int b = 0; b < KB; b++) // the keys' registers are free now; the values arrive while the keys leave
        {
            const int i = 64 * b + lane;
            if (i < mine) val[b] = vin_w[i];
        }
        __syncthreads();
        uint32_t dpack[KB / 4 > 0 ? KB / 4 : 1]; // the digits of the KB positions this thread writes out, for the values' turn
#pragma unroll
        for (int j = 0; j < KB; j++)
        {
            const int p = t + 256 * j;
            if ((j & 3) == 0) dpack[j >> 2] = 0u;
            if (p < count)
            {
                const uint32_t k = stage_k[p], d = (k >> shift) & mask;
                dpack[j >> 2] |= d << (8 * (j & 3));
                kout[(int64_t)gdelta[d] + p] = k;
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < KB; b++)
            if (64 * b + lane < mine) stage_k[rk[b]] = val[b];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KB; j++)
        {
            const int p = t + 256 * j;
            if (p < count) vout[(int64_t)gdelta[(dpack[j >> 2] >> (8 * (j & 3))) & 0xFFu] + p] = stage_k[p];
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < KB; b++)
        if (64 * b + lane < mine)
        {
            const uint32_t p = wcnt[wave][(key[b] >> shift) & mask] + rk[b];
            stage_k[p] = key[b];
            stage_v[p] = val[b];
        }
    __syncthreads();
    for (int p = t; p < count; p += 256)
    {
        const uint32_t k = stage_k[p];
        const int64_t dst = (int64_t)gdelta[(k >> shift) & mask] + p;
        kout[dst] = k;
        vout[dst] = stage_v[p];
    }
    if (CENSUS && chunk == 0) publish_census(census, r.chunks, t);
}
template <bool IDENTITY_VALUES, int CH, bool DIRECT, bool CENSUS = false>
__global__ void __launch_bounds__(256) rs_scatter_kernel(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                          uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, int64_t n,
                                                          const unsigned long long *n_dev, int shift, int nbits, RadixScratchView r, const uint32_t *skip_flag,
                                                          const uint32_t *__restrict__ acc, uint32_t *__restrict__ acc_clear, DepthCensus census = DepthCensus{})
{
    rs_scatter_body<IDENTITY_VALUES, CH, DIRECT, CENSUS, false>(kin, vin, kout, vout, n, n_dev, shift, nbits, r, skip_flag, acc, acc_clear, census);
}
// the TWO_PHASE flavour with the register budget of five workgroups per CU (1280 resident chunks = 5.2 M pairs)
template <int CH>
__global__ void __launch_bounds__(256, 5) rs_scatter_two_phase_kernel(const uint32_t *__restrict__ kin, const uint32_t *__restrict__ vin,
                                                                       uint32_t *__restrict__ kout, uint32_t *__restrict__ vout, int64_t n,
                                                                       const unsigned long long *n_dev, int shift, int nbits, RadixScratchView r,
                                                                       const uint32_t *skip_flag, const uint32_t *__restrict__ acc, uint32_t *__restrict__ acc_clear)
{
    rs_scatter_body<false, CH, true, false, true>(kin, vin, kout, vout, n, n_dev, shift, nbits, r, skip_flag, acc, acc_clear, DepthCensus{});
}

// The chunk length is a run-time property of the sort's scratch (ts2d_common.h: 1024 / 2048 / 4096 pairs); every launcher instantiates its kernel
// for the three of them.  Inside the braces CH is the compile-time length.
#define TS_WITH_CHUNK(chunk, ...)                                    \
    switch (chunk)                                                   \
    {                                                                \
    case TS_RS_CHUNK_SMALL: { constexpr int CH = TS_RS_CHUNK_SMALL; __VA_ARGS__; } break; \
    case TS_RS_CHUNK_MID: { constexpr int CH = TS_RS_CHUNK_MID; __VA_ARGS__; } break;     \
    default: { constexpr int CH = TS_RS_CHUNK; __VA_ARGS__; } break; \
    }
void radix_hist(const uint32_t *kin, int64_t n, const unsigned long long *n_dev, int shift, int nbits, const RadixScratchView &r, hipStream_t s,
                const DepthCensus *census = nullptr, const uint32_t *skip_flag = nullptr)
{
    const dim3 grid((unsigned)r.chunks);
    const uint32_t mask = (1u << nbits) - 1u;
    TS_WITH_CHUNK(r.chunk,
                  if (census) hipLaunchKernelGGL((rs_hist_kernel<true, CH>), grid, dim3(256), 0, s, kin, n, n_dev, shift, mask, r, *census, skip_flag);
                  else hipLaunchKernelGGL((rs_hist_kernel<false, CH>), grid, dim3(256), 0, s, kin, n, n_dev, shift, mask, r, DepthCensus{}, skip_flag))
}
template <int CH>
void radix_scatter_ch(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t n, const unsigned long long *n_dev, int shift,
                      int nbits, const RadixScratchView &r, hipStream_t s, const uint32_t *skip_flag, const uint32_t *acc, uint32_t *acc_clear)
{
    const dim3 grid((unsigned)r.chunks);
#define TS_SCATTER(ID, D) hipLaunchKernelGGL((rs_scatter_kernel<ID, CH, D>), grid, dim3(256), 0, s, kin, vin, kout, vout, n, n_dev, shift, nbits, r, skip_flag, acc, acc_clear)
    if (acc && vin)
    {
        if constexpr (CH == TS_RS_CHUNK) // the instance sort of large scenes: keys and values through one staging array (rs_scatter_body)
            hipLaunchKernelGGL((rs_scatter_two_phase_kernel<CH>), grid, dim3(256), 0, s, kin, vin, kout, vout, n, n_dev, shift, nbits, r, skip_flag, acc, acc_clear);
        else TS_SCATTER(false, true);
    }
    else if (acc) TS_SCATTER(true, true);
    else if (vin) TS_SCATTER(false, false);
    else TS_SCATTER(true, false);
#undef TS_SCATTER
}
void radix_scatter(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t n, const unsigned long long *n_dev, int shift,
                   int nbits, const RadixScratchView &r, hipStream_t s, const uint32_t *skip_flag = nullptr, const uint32_t *acc = nullptr,
                   uint32_t *acc_clear = nullptr)
{
    TS_WITH_CHUNK(r.chunk, radix_scatter_ch<CH>(kin, vin, kout, vout, n, n_dev, shift, nbits, r, s, skip_flag, acc, acc_clear))
}
void radix_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t n, const unsigned long long *n_dev, int shift,
                int nbits, const RadixScratchView &r, hipStream_t s, const uint32_t *skip_flag = nullptr)
{
    radix_hist(kin, n, n_dev, shift, nbits, r, s, nullptr, skip_flag);
    radix_scatter(kin, vin, kout, vout, n, n_dev, shift, nbits, r, s, skip_flag);
}
// The ticket-free pass: the histogram adds into slabacc[which] (cleared by whoever ran before), the scatter reads it and clears the other
// buffer for the pass after this one.  Every block reads the totals of all slabs, so sorts of more than TS_DIRECT_MAX_SLABS slabs (12.6 M
// pairs at 4096 per chunk) keep the hierarchical pass, whose cost does not grow with the slab count.
// g_force_tickets: a lab-library switch (csrc/ts2d_lab.h, ts2d_lab_force_ticket_passes; the symbol is not exported and nothing in the product
// library sets it): every sort takes the hierarchical passes that otherwise only sorts of more than 48 slabs reach (> 6.3 M triangles, > 12.6 M
// instances), so that the suite executes them -- including the ticket-path census that produces num_rendered there.
bool g_force_tickets = false;
bool g_force_pass4 = false; // same kind of switch (ts2d_lab_force_depth_pass4): the depth sort never skips its fourth pass
bool radix_direct_ok(const RadixScratchView &r) { return !g_force_tickets && r.slabs <= TS_DIRECT_MAX_SLABS; }
void radix_pass_direct(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, int64_t n, const unsigned long long *n_dev, int shift,
                       int nbits, const RadixScratchView &r, int which, hipStream_t s, const uint32_t *skip_flag = nullptr)
{
    const dim3 grid((unsigned)r.chunks);
    const uint32_t mask = (1u << nbits) - 1u;
    TS_WITH_CHUNK(r.chunk, hipLaunchKernelGGL((rs_hist_direct_kernel<CH>), grid, dim3(256), 0, s, kin, n, n_dev, shift, mask, r, r.slabacc[which], skip_flag))
    radix_scatter(kin, vin, kout, vout, n, n_dev, shift, nbits, r, s, skip_flag, r.slabacc[which], r.slabacc[which ^ 1]);
}

// ---- step 2: tiles_sorted = tiles_touched[perm], 64-bit block sums, their prefix, N ---------------------------------------
constexpr int SB = 1024; // triangles per scan block (256 threads x 4)

// The block sums stay RAW and nobody waits for a last block (round 3): every emission block adds up the sums in front of it itself
// (blocksum[nblocks] = N is already there from the depth sort's census): up to 2048 of them (2 M triangles), eight loads per thread.  Beyond
// that (round 4) a second level `supersum` holds the sum of every 64 consecutive block sums (fire-and-forget 64-bit atomics, zeroed by the
// step's first launch), so an emission block adds at most nblocks / 64 + 63 values at any size.  Until then such scenes fell back to an
// elected block that turned the sums into their prefix -- one same-address ticket atomic per block, 4883 in a row at 5 M triangles: scan
// 0.113 -> 0.087 ms there.  (Below 2048 blocks the atomics cost more than they save: 13 -> 18 us at 1 M triangles, hence the two forms.)
__global__ void __launch_bounds__(256) gather_blocksum_kernel(int P, GeometryStateView g, bool two_level)
{
    __shared__ unsigned long long wsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i0 = blockIdx.x * SB + 4 * t;
    const uint32_t *ids = sorted_ids(g);
    uint32_t v[4];
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        v[k] = (i0 + k < P) ? g.tiles_touched[ids[i0 + k]] : 0u;
        sum += v[k];
    }
    if (i0 + 3 < P) *(uint4 *)(g.tiles_sorted + i0) = make_uint4(v[0], v[1], v[2], v[3]);
    else
        for (int k = 0; k < 4; k++)
            if (i0 + k < P) g.tiles_sorted[i0 + k] = v[k];
    // block sum: tile counts are < 2^32 each, a block's sum may not be
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) wsum[wave] = sum;
    __syncthreads();
    if (t == 0)
    {
        const unsigned long long total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        g.blocksum[blockIdx.x] = total;
        if (two_level && total) __hip_atomic_fetch_add((unsigned long long *)g.supersum + (blockIdx.x >> 6), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- step 3: instance slots + emission ------------------------------------------------------------------------------------------
// One lane per depth-ordered triangle.  Triangles covering up to SMALL tiles are emitted by their own lane; larger ones
// (stress scenes where a triangle spans thousands of tiles) are emitted cooperatively by the whole wave so that a single
// lane never serialises a long loop.  Tiles of one triangle are emitted row-major like the reference's loop
// (rasterizer.cu:63-73); the later sort is by tile id, so only the order BETWEEN triangles matters.
constexpr uint32_t SMALL = 32;

// Round 5, measured on three scenes with the five combinations alternating on one box (profiles/r05_emission_variants.txt; tools/r05_call4.sh):
// a register budget for 6 waves per SIMD (80 registers, 3 spilled; the compiler's own choice is 94 = 5 waves) is worth 0-3 %, and requesting
// the rectangle / record gather EARLY, under the block sums and the scan, costs 10 % (1 M triangles: 0.054 -> 0.064 ms; 5 M: 0.245 -> 0.277):
// the gather's 96-128 bytes per lane sit in registers across the scan and the loads queue in front of the block sums the scan waits for.
#ifndef TS_EMIT_WAVES // register budget for N waves per SIMD (0: the compiler's choice)
#define TS_EMIT_WAVES 6
#endif
#ifndef TS_EMIT_PREFETCH
#define TS_EMIT_PREFETCH 0
#endif
#if TS_EMIT_WAVES > 0
__global__ void __launch_bounds__(256, TS_EMIT_WAVES) scan_emit_kernel(
#else
__global__ void __launch_bounds__(256) scan_emit_kernel(
#endif
int P, int grid_x, int ntiles, GeometryStateView g, BinningStateView b, uint2 *ranges,
                                                         float *contrib_sum, float *contrib_max, long long capacity, int32_t *status, bool two_level,
                                                         QuadMaskArgs qmask)
{
    __shared__ uint32_t wtot[4];
    __shared__ unsigned long long wpart[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = blockIdx.x * 256 + t;
    // output clears that used to be three memset launches: tile ranges (rasterizer.cu:223) and the contribution statistics
    for (int k = i; k < ntiles; k += gridDim.x * 256) ranges[k] = make_uint2(0u, 0u);
    if (b.rs.tickets)
    {
        for (int k = i; k < b.rs.slabs + TS_RS_TICKET_EXTRA; k += gridDim.x * 256) b.rs.tickets[k] = 0u; // the tile sort's tickets
        for (int k = i; k < b.rs.slabs * NB; k += gridDim.x * 256) b.rs.slabacc[0][k] = 0u; // ... and its first pass's slab totals
    }
    if (contrib_sum && i < P)
    {
        contrib_sum[i] = 0.0f;
        contrib_max[i] = 0.0f;
    }
    const bool valid = i < P;
    uint32_t tiles = valid ? g.tiles_sorted[i] : 0u;
    const uint32_t id_ahead = valid ? sorted_ids(g)[i] : 0u; // wanted after the scan: requested now, one round trip less behind it
    // (-DTS_EMIT_PREFETCH=1: what hangs on it -- the tile rectangle and the head of the render record -- requested as soon as the id is there; a
    // measured negative, see above)
    uint2 rect = {0u, 0u};
    float4 rec0 = make_float4(0, 0, 0, 0), rec1 = rec0, rec2 = rec0;
#if TS_EMIT_PREFETCH
    if (tiles > 0)
    {
        rect = g.rect[id_ahead];
        const float4 *rp = g.rec + 4 * (size_t)id_ahead;
        rec0 = rp[0];
        rec1 = rp[1];
        if (qmask.variant == 3) rec2 = rp[2];
    }
#endif
    // Everything in front of this block, requested together and reduced once: the earlier quarters of this scan block (scan blocks are 1024
    // triangles = four of these 256-lane blocks), the raw sums of the scan blocks of its group of 64, the group sums in front of that.
    const int sblock = blockIdx.x >> 2, quarter = blockIdx.x & 3;
    unsigned long long part = 0;
    if (two_level)
    {
        for (int k = t; k < (sblock >> 6); k += 256) part += g.supersum[k];
        if (t < (sblock & 63)) part += g.blocksum[(sblock & ~63) + t];
    }
    else
        for (int k = t; k < sblock; k += 256) part += g.blocksum[k];
    for (int q = 0; q < quarter; q++)
    {
        const int j = (sblock * 4 + q) * 256 + t;
        part += (j < P) ? g.tiles_sorted[j] : 0u;
    }
    if (capacity >= 0) // sync-free forward: the instance count is only known here; over capacity nothing is emitted
    {
        const unsigned long long live = g.blocksum[(P + SB - 1) / SB];
        const bool over = live > (unsigned long long)capacity;
        if (i == 0 && status) *status = over ? 1 : 0;
        if (over) tiles = 0u;
    }
    else if (i == 0 && status) *status = 0; // the host knows the count (the reference's sequence, or the exact re-run after an overflow)
    // inclusive prefix inside the block (wave64 DPP scan + the three preceding waves' totals) on top of what lies in front of the block
    const uint32_t inc = wave_inclusive_scan(tiles, lane);
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 63) wtot[wave] = inc;
    if (lane == 0) wpart[wave] = part;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += wtot[w];
    const unsigned long long qbase = wpart[0] + wpart[1] + wpart[2] + wpart[3];
    const uint32_t incl = (uint32_t)(qbase + before + inc); // N < 2^31 is checked on the host before anything is emitted
    if (valid) g.offsets[i] = incl;
    const uint32_t id = tiles > 0 ? id_ahead : 0u;
    const uint32_t off = incl - tiles; // exclusive prefix
#if !TS_EMIT_PREFETCH
    if (tiles > 0)
    {
        rect = g.rect[id];
        const float4 *rp = g.rec + 4 * (size_t)id;
        rec0 = rp[0];
        rec1 = rp[1];
        if (qmask.variant == 3) rec2 = rp[2];
    }
#endif
    const uint32_t minx = rect.x & 0xffffu, miny = rect.x >> 16, maxx = rect.y & 0xffffu, maxy = rect.y >> 16;
    uint32_t *tile_out = b.k[0], *val_out = b.v[0];
    // the four spare bits of an instance's value say which 8x8 quadrants of its tile the triangle's support can reach (ts2d_support.h; both
    // variants since round 5, ts2d_common.h: QuadMaskArgs); the blend kernels' quadrant waves then skip the other entries unseen
    constexpr bool qm = true;
    const float quad_g2 = qmask.g2;
    auto setup_from = [&](const float4 &r0, const float4 &r1, const float4 &r2, uint32_t tminx, uint32_t tminy, uint32_t tmaxx, uint32_t tmaxy) {
        if (qmask.variant == 0) return quad_setup_all(); // lab library only: every quadrant (ts2d_lab_force_all_quadrants)
        if (qmask.variant == 3) // the head of the triangle's record and its tile rectangle
        {
            const float E = quad_g2 == 2.0f ? support_scale<true>(1.0f, quad_g2) : support_scale<false>(1.0f, quad_g2);
            return quad_setup_3d(r0, r1, r2, E, qmask.tan_fovx, qmask.tan_fovy, qmask.W, qmask.H, qmask.inv_W, qmask.inv_H, (float)(tminx * TS_TILE) - 1.0f,
                                 (float)(tminy * TS_TILE) - 1.0f, (float)(tmaxx * TS_TILE), (float)(tmaxy * TS_TILE));
        }
        const float E = quad_g2 == 2.0f ? support_scale<true>(r1.z, quad_g2) : support_scale<false>(r1.z, quad_g2);
        return quad_setup(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, E);
    };
    auto setup_of = [&](uint32_t tri, uint32_t tminx, uint32_t tminy, uint32_t tmaxx, uint32_t tmaxy) { // another lane's triangle: gathered here
        const float4 *rp = g.rec + 4 * (size_t)tri;
        return setup_from(rp[0], rp[1], qmask.variant == 3 ? rp[2] : make_float4(0, 0, 0, 0), tminx, tminy, tmaxx, tmaxy);
    };
    QuadSetup qs{};
    // The block's instances are one contiguous run of the list.  A lane writing its triangle's few slots straight to memory issues 4-byte
    // stores a few slots apart from its neighbours' (a 32-64 byte fabric write each on this chip); runs of up to STAGE instances are put
    // together in LDS instead and leave as coalesced rows.
    constexpr uint32_t STAGE = 2048;
    // stage_t: a staged instance's tile; region B: its value (plain staging) or -- when the masks are formed -- the 256 triangles' affine mask
    // constants (ts2d_support.h: QuadAffine, 64 bytes each), in which case a staged instance is (dx | dy << 12 | triangle slot << 24)
    __shared__ uint32_t stage_t[STAGE];
    __shared__ __attribute__((aligned(16))) float4 region_b[256 * 4];
    uint32_t *const stage_v = (uint32_t *)region_b;
    const uint32_t run0 = (uint32_t)qbase, run = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    const bool staged = run <= STAGE;
    // Staged runs form their masks in the flush loop below, one lane per INSTANCE, evenly spread over the block (a lane that walks its triangle's
    // tiles makes the whole wave wait for the triangle with the most tiles).  Round 4 gathered the triangle's record and redid the whole setup
    // per instance (~130 VALU instructions + a 32-byte gather each: 0.052 ms against 0.029 without masks at 1 M triangles); round 5 does the
    // setup once per TRIANGLE, leaves its 16 affine constants in LDS, and an instance costs two FMAs per edge + the compares.  Runs too long
    // for the stage (big triangles) test per tile where they write.
    const bool qstage = qm && staged;
    const uint32_t rw = maxx - minx, rh = maxy - miny;
    if (qm && tiles > 0)
    {
        qs = setup_from(rec0, rec1, rec2, minx, miny, maxx, maxy);
        if (qstage)
        {
            const QuadAffine qa = quad_anchor(qs, id, minx, miny, rw, rh);
            region_b[4 * t] = qa.a; region_b[4 * t + 1] = qa.b; region_b[4 * t + 2] = qa.c; region_b[4 * t + 3] = qa.d;
        }
    }
    if (tiles > 0 && tiles <= SMALL)
    {
        uint32_t o = off;
        if (staged)
        {
            o -= run0;
            for (uint32_t y = miny; y < maxy; y++)
                for (uint32_t x = minx; x < maxx; x++)
                {
                    stage_t[o] = qstage ? ((x - minx) | ((y - miny) << 12) | ((uint32_t)t << 24)) : y * grid_x + x;
                    if (!qstage) stage_v[o] = id;
                    o++;
                }
        }
        else
            for (uint32_t y = miny; y < maxy; y++)
                for (uint32_t x = minx; x < maxx; x++)
                {
                    tile_out[o] = y * grid_x + x;
                    val_out[o] = qm ? id | (quadrant_mask(qs, (float)(x * TS_TILE), (float)(y * TS_TILE)) << TS_ID_BITS) : id;
                    o++;
                }
    }
    unsigned long long big = ballot64(tiles > SMALL);
    while (big)
    {
        const int j = __builtin_ctzll(big);
        big &= big - 1;
        const uint32_t t_minx = __shfl(minx, j), t_miny = __shfl(miny, j), t_maxx = __shfl(maxx, j), t_maxy = __shfl(maxy, j);
        const uint32_t t_tiles = __shfl(tiles, j), t_off = __shfl(off, j), t_id = __shfl(id, j);
        const uint32_t w = t_maxx - t_minx;
        QuadSetup tq{};
        if (qm && !staged) tq = setup_of(t_id, t_minx, t_miny, t_maxx, t_maxy); // every lane of the wave for itself: the same record, no 20-value broadcast
        for (uint32_t k = lane; k < t_tiles; k += 64)
        {
            const uint32_t y = t_miny + k / w, x = t_minx + k % w;
            if (staged)
            {
                stage_t[t_off - run0 + k] = qstage ? ((k % w) | ((k / w) << 12) | ((uint32_t)(wave * 64 + j) << 24)) : y * grid_x + x;
                if (!qstage) stage_v[t_off - run0 + k] = t_id;
            }
            else
            {
                tile_out[t_off + k] = y * grid_x + x;
                val_out[t_off + k] = qm ? t_id | (quadrant_mask(tq, (float)(x * TS_TILE), (float)(y * TS_TILE)) << TS_ID_BITS) : t_id;
            }
        }
    }
    if (staged)
    {
        __syncthreads();
        for (uint32_t k = t; k < run; k += 256)
        {
            uint32_t tl = stage_t[k], v;
            if (qstage)
            {
                const uint32_t dx = tl & 0xfffu, dy = (tl >> 12) & 0xfffu, slot = tl >> 24;
                QuadAffine qa;
                qa.a = region_b[4 * slot]; qa.b = region_b[4 * slot + 1]; qa.c = region_b[4 * slot + 2]; qa.d = region_b[4 * slot + 3];
                const uint32_t org = __float_as_uint(qa.d.z), x = (org & 0xffffu) + dx, y = (org >> 16) + dy;
                v = __float_as_uint(qa.d.y) | (quadrant_mask_affine(qa, dx, dy, x, y) << TS_ID_BITS);
                tl = y * grid_x + x;
            }
            else v = stage_v[k];
            tile_out[run0 + k] = tl;
            val_out[run0 + k] = v;
        }
    }
}

// Four consecutive instances per thread (one dwordx4 + the word in front): a quarter of the workgroups and of the loads of the one-key-per-thread
// form (10.4 -> 7.6 us at the headline, 38.9 -> 19.1 us at 5 M triangles: profiles/r05_notes.md section 12).  `tile` is 16-byte aligned (ts_carve); words past N are never looked at.
__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t N, const unsigned long long *n_dev, const uint32_t *__restrict__ tile,
                                                           uint2 *__restrict__ ranges)
{
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (n_dev)
    {
        const unsigned long long live = *n_dev;
        N = (live <= (unsigned long long)N) ? (int64_t)live : 0;
    }
    if (i0 >= N) return;
    uint32_t k[4];
    if (i0 + 3 < N)
    {
        const uint4 q = *(const uint4 *)(tile + i0);
        k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
    }
    else
        for (int j = 0; j < 4; j++) k[j] = i0 + j < N ? tile[i0 + j] : 0u;
    uint32_t prev = i0 > 0 ? tile[i0 - 1] : 0u;
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
        const int64_t i = i0 + j;
        if (i >= N) break;
        const uint32_t cur = k[j];
        if (i == 0) ranges[cur].x = 0;
        else if (cur != prev)
        {
            ranges[prev].y = (uint32_t)i;
            ranges[cur].x = (uint32_t)i;
        }
        if (i == N - 1) ranges[cur].y = (uint32_t)N;
        prev = cur;
    }
}

__global__ void zero_words_kernel(uint32_t *p, size_t n) // 64-bit count and index: 16 P words at P near 2^28 pass 2^31
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
// ---- small scenes: the whole depth order in ONE launch ------------------------------------------------------------------------------
// Up to TS_DEPTH_SMALL_MAX triangles the eight launches of the depth sort and the launch of the block sums order a few thousand keys in 4-9 us
// each -- dependent launches, each one a load -> LDS -> barrier -> store chain with a handful of workgroups on the chip (BASELINE configs[0],
// 10 k triangles: 47 + 7 of the step's 166 us).  Here ONE workgroup of 16 waves keeps the (key, id) pairs in registers (wave w: a contiguous
// range of `per` <= 1024 positions, 64 consecutive pairs per step), ranks them with the same wave-local match as rs_scatter_body, exchanges
// them through LDS after every pass, and then does what the census, publish_census and gather_blocksum_kernel do: N to the device word and
// the pinned host word, tiles_sorted, the raw block sums.  Same passes (the fourth skipped under the same condition), same stable order,
// so sv[1] holds exactly the ids the multi-launch path leaves in sorted_ids(); top_const stays 0 (the order is always in sk[1] / sv[1]).
constexpr int TS_DEPTH_SMALL_MAX = 12288; // the kernel's capacity (level with the LSD passes of 1024-pair chunks at ~12 k triangles: 34 us either way); used up to TS_DEPTH_SMALL_USE
constexpr int DS_WAVES = 16, DS_KB = TS_DEPTH_SMALL_MAX / (64 * DS_WAVES);
// The one-launch form lives on gfx950's 160 KB of LDS per workgroup (two stages of TS_DEPTH_SMALL_MAX words + the per-wave digit table): 64 KB
// parts cannot hold it, and nothing else in the library may quietly push it over
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "depth_order_small_kernel is sized for gfx950 (160 KB LDS per workgroup)"
#endif
static_assert(2 * TS_DEPTH_SMALL_MAX * 4 + DS_WAVES * NB * 4 + DS_WAVES * 8 + (TS_DEPTH_SMALL_MAX / SB) * 8 + 2 * DS_WAVES * 4 + 16 <= 160 * 1024 - 8 * 1024,
              "depth_order_small_kernel: static LDS must leave 8 KB of the 160 KB for the runtime / alignment");
__global__ void __launch_bounds__(64 * DS_WAVES) depth_order_small_kernel(int P, GeometryStateView g, unsigned long long *host_out)
{
    __shared__ __attribute__((aligned(16))) uint32_t stage_k[TS_DEPTH_SMALL_MAX], stage_v[TS_DEPTH_SMALL_MAX];
    __shared__ uint32_t wcnt[DS_WAVES][NB]; // per-wave digit counts, then the start of the (wave, digit) run
    __shared__ uint32_t wtot[4];
    __shared__ unsigned long long csum[DS_WAVES], bsum[TS_DEPTH_SMALL_MAX / SB];
    __shared__ uint32_t cor[DS_WAVES], cnand[DS_WAVES];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int per = ((P + 64 * DS_WAVES - 1) / (64 * DS_WAVES)) * 64; // positions per wave
    const int base = wave * per;
    const int mine = P - base < per ? (P - base > 0 ? P - base : 0) : per;
    const uint32_t *keys = (const uint32_t *)g.depth;
    uint32_t key[DS_KB], val[DS_KB], rk[DS_KB];
    unsigned long long tsum = 0;
    uint32_t kor = 0u, knand = 0u;
    if (t < TS_DEPTH_SMALL_MAX / SB) bsum[t] = 0ull;
#pragma unroll
    for (int b = 0; b < DS_KB; b++) // branch-free: the register arrays stay scalars the compiler can place one by one
    {
        const int i = 64 * b + lane;
        const bool in = i < mine;
        const int src = in ? base + i : 0;
        const uint32_t k = keys[src], tt = g.tiles_touched[src];
        key[b] = in ? k : 0xFFFFFFFFu;
        val[b] = (uint32_t)src;
        tsum += in ? tt : 0u;
        const bool vis = in && k != 0u; // culled triangles (key 0) do not count: rs_hist_census_direct_kernel
        kor |= vis ? k : 0u;
        knand |= vis ? ~k : 0u;
    }
    for (int o = 32; o > 0; o >>= 1)
    {
        tsum += __shfl_xor(tsum, o);
        kor |= __shfl_xor(kor, o);
        knand |= __shfl_xor(knand, o);
    }
    if (lane == 0) { csum[wave] = tsum; cor[wave] = kor; cnand[wave] = knand; }
    __syncthreads();
    unsigned long long N = 0;
    uint32_t o_all = 0u, n_all = 0u;
#pragma unroll
    for (int w = 0; w < DS_WAVES; w++) { N += csum[w]; o_all |= cor[w]; n_all |= cnand[w]; }
    const uint32_t varying = o_all ^ ~n_all;             // publish_census: or ^ and, and = ~(or of the complements)
    const int passes = (varying >> 24) == 0u ? 3 : 4;    // no visible triangle: or = 0, and = ~0 -> four passes, like the flag there
    uint32_t *cnt = wcnt[wave];
#pragma nounroll
    for (int pass = 0; pass < passes; pass++)
    {
        const int shift = 8 * pass;
#pragma unroll
        for (int k = 0; k < NB / 64; k++) cnt[lane + 64 * k] = 0u;
        wave_lds_order();
#pragma unroll
        for (int b = 0; b < DS_KB; b++)
        {
            if (64 * b >= per) continue; // wave-uniform
            const bool valid = 64 * b + lane < mine;
            const uint32_t d = (key[b] >> shift) & 0xFFu;
            // lanes whose digit differs from mine in some bit: (ballot of bit i) xor (my bit i, sign-extended), or-ed over the bits -- two
            // 32-bit halves, three instructions per bit and half (the select form of rs_scatter_body costs ~100 instructions per step; this
            // kernel runs on ONE compute unit and is bound by exactly these)
            const unsigned long long vm = ballot64(valid);
            uint32_t mis_lo = ~(uint32_t)vm, mis_hi = ~(uint32_t)(vm >> 32);
#pragma unroll
            for (int bit = 0; bit < 8; bit++)
            {
                const unsigned long long bb = ballot64((d >> bit) & 1u);
                const uint32_t e = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit, 1); // all ones when my bit is set
                mis_lo |= (uint32_t)bb ^ e;
                mis_hi |= (uint32_t)(bb >> 32) ^ e;
            }
            const uint32_t m_lo = ~mis_lo, m_hi = ~mis_hi; // valid lanes with my digit
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            const uint32_t c = (uint32_t)(__popc(m_lo) + __popc(m_hi));
            uint32_t seen = 0;
            if (valid) seen = cnt[d];
            wave_lds_order(); // every lane has read its digit's count before the group leaders advance it
            if (valid && rank == c - 1u) cnt[d] = seen + c;
            wave_lds_order();
            rk[b] = seen + rank;
        }
        __syncthreads();
        uint32_t c[DS_WAVES], tot = 0u, inc = 0u;
        if (t < NB) // thread d = digit d
        {
#pragma unroll
            for (int w = 0; w < DS_WAVES; w++) { c[w] = wcnt[w][t]; tot += c[w]; }
            inc = wave_inclusive_scan(tot, lane);
            if (lane == 63) wtot[wave] = inc;
        }
        __syncthreads();
        if (t < NB)
        {
            uint32_t run = inc - tot;
            for (int w = 0; w < wave; w++) run += wtot[w];
#pragma unroll
            for (int w = 0; w < DS_WAVES; w++) { wcnt[w][t] = run; run += c[w]; }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < DS_KB; b++)
            if (64 * b + lane < mine)
            {
                const uint32_t p = cnt[(key[b] >> shift) & 0xFFu] + rk[b];
                stage_k[p] = key[b];
                stage_v[p] = val[b];
            }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < DS_KB; b++) // (positions past `mine` hold stale pairs that nothing uses)
        {
            key[b] = stage_k[base + 64 * b + lane];
            val[b] = stage_v[base + 64 * b + lane];
        }
    }
    // the order, the tile counts in that order, their raw sums per scan block (64 consecutive positions never straddle a block of SB)
#pragma unroll
    for (int b = 0; b < DS_KB; b++)
    {
        if (64 * b >= per) continue;
        const int i = base + 64 * b + lane;
        unsigned long long tiles = 0;
        if (64 * b + lane < mine)
        {
            const uint32_t tt = g.tiles_touched[val[b]];
            g.sk[1][i] = key[b];
            g.sv[1][i] = val[b];
            g.tiles_sorted[i] = tt;
            tiles = tt;
        }
        for (int o = 32; o > 0; o >>= 1) tiles += __shfl_xor(tiles, o);
        if (lane == 0 && tiles) atomicAdd(&bsum[(base + 64 * b) / SB], tiles);
    }
    __syncthreads();
    const int nblocks = (P + SB - 1) / SB;
    if (t < nblocks) g.blocksum[t] = bsum[t];
    if (t == 0)
    {
        g.blocksum[nblocks] = N; // = DepthCensus::n_out
        *g.top_const = 0u;
        if (host_out) __hip_atomic_store(host_out, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- mid-sized scenes: the depth order by SAMPLED SPLITTERS + per-bucket sorts in LDS (round 6) -------------------------------------------------
// Above TS_DEPTH_SMALL_MAX triangles the LSD sort above is eight dependent launches (census histogram, four scatters, three histograms; the fourth
// pair returns at once on most scenes) + the launch of the block sums: 57 us for 0.7 MB of pairs at 93 k triangles, 59 at 300 k, 75 at 1 M --
// launch and load -> LDS -> store latency, not bytes.  Four launches instead:
//   K0  depth_split_sample_kernel   4096 keys taken at equal index strides (1024 below 150 k triangles: -2 us, and the buckets are small against K3's capacity
//                                   whatever their balance); workgroup b finds splitter b (the sample of rank 16 b + 15) by a radix select.
//                                   (Splitters from a SAMPLE, not from the key range: the buckets hold P / 256 +- 25 % pairs whatever the depth
//                                   distribution is -- a far background, an object that fills one octave -- where equal slices of the bit range
//                                   would put most of the scene into a few of them.)
//   K1  depth_split_hist_kernel     bucket(key) = number of splitters below the key (8-step search in LDS) -> a byte per key, the chunk's bucket
//                                   counts (table + per-slab totals, like rs_hist_direct_kernel), the chunk's part of N
//   K2  depth_split_scatter_kernel  the stable scatter of rs_scatter_body<IDENTITY, DIRECT> on those bytes: (key, id) pairs bucket by bucket in
//                                   sk[0] / sv[0]; chunk 0 leaves the 256 bucket starts and publishes N (device word + pinned host word)
//   K3  depth_bucket_sort_kernel    one workgroup per bucket (4 waves below 300 k triangles, 16 above): its pairs in registers, a stable LSD sort in
//                                   LDS on the bits of (key - smallest visible key of the bucket) that vary inside the bucket (13 of them = two
//                                   passes, 8 + 5 ballots, where the full key needs three or four of 8), then what gather_blocksum_kernel did:
//                                   tiles_sorted and the raw block sums (64-bit atomics; the sums were cleared by K0).
// bucket() is monotone in the key and K2 is stable, K3 is a stable sort by the full key inside a bucket: sk[1] / sv[1] hold exactly the (key, id)
// order of the LSD passes -- checked form against form in tests/test_parity_gpu.py::test_split_depth_order_equals_the_multi_launch_forms.  A bucket
// larger than K3's registers (16 384 pairs: many equal or nearly equal depths) is sorted by the same workgroup through global memory, tile by tile --
// slow, correct.
// Measured against the LSD passes alternating on one box (profiles/r06_depth_split.txt): 93 k triangles 57 -> 43 us (step 0.335 -> 0.321 ms), 300 k
// 59 -> 47 (0.613 -> 0.605), 1 M 75 -> 65 us of kernels and NO difference in the step (the largest bucket holds 7 500 pairs = 8 steps x 2 passes of
// a ranking that is issue-bound with 16 waves on the compute unit: K3 takes 35 us there) -- hence the switch-over below.
#ifndef TS_DEPTH_SPLIT_MAX_VALUE // (variant builds: tools/build_obj_variant.sh ... binning "-DTS_DEPTH_SPLIT_MAX_VALUE=1600000")
#define TS_DEPTH_SPLIT_MAX_VALUE 500000
#endif
constexpr int TS_DEPTH_SPLIT_MAX = TS_DEPTH_SPLIT_MAX_VALUE; // the product's switch-over: measured level with the LSD passes at 1 M triangles, ahead below
constexpr int TS_DEPTH_SPLIT_HARD_MAX = 1600000; // what the form supports (buckets of P / 256 pairs on average against DB_CAP = 16384): lab library, mode 2
constexpr int DSPL_SAMPLES = 4096, DSPL_SAMPLES_SMALL = 1024, DSPL_SMALL_BELOW = 150000; // below: P / 256 < 600 pairs per bucket against K3's 4096
constexpr int DB_WAVES = 16, DB_KB = 16, DB_CAP = 64 * DB_WAVES * DB_KB; // the large form of depth_bucket_sort_kernel
#ifndef TS_DB_SMALL_BELOW_VALUE
#define TS_DB_SMALL_BELOW_VALUE 300000
#endif
constexpr int DB_SMALL_WAVES = 4, DB_SMALL_BELOW = TS_DB_SMALL_BELOW_VALUE; // below: buckets of P / 256 < 1200 pairs on average against 4096
int g_depth_split_mode = 0; // lab library: 0 = by size, 1 = never (the LSD passes), 2 = up to TS_DEPTH_SPLIT_HARD_MAX
int g_depth_bucket_cap = DB_CAP; // lab library: a smaller register capacity sends ordinary buckets through K3's global-memory path

struct DepthSplit
{
    uint32_t *splitters;           // 256 words: splitter b, b < 255; [255] = ~0
    uint32_t *bucket_start;        // 256 words
    unsigned long long *chunk_sum; // per chunk: sum of tiles_touched
    uint8_t *digit;                // P bytes (lives in sv[1] until K3 overwrites it)
};
__host__ __device__ inline DepthSplit depth_split_of(const GeometryStateView &g)
{
    // scratch inside the state: `offsets` is written by scan_emit_kernel, after K3 (P > TS_DEPTH_SMALL_MAX words: room for 1024 + 2 chunks)
    DepthSplit d;
    d.splitters = g.offsets;
    d.bucket_start = g.offsets + NB;
    d.chunk_sum = (unsigned long long *)(g.offsets + 1024);
    d.digit = (uint8_t *)g.sv[1];
    return d;
}

// Depth keys are positive floats' bit patterns, and 0 for culled triangles.  Inside K0 and K3 the keys are taken relative to the smallest VISIBLE key
// (monotone; culled stay 0): the bytes that all visible keys share -- and that 0 does not -- then cost no pass.
__device__ __forceinline__ uint32_t depth_key_adjust(uint32_t key, uint32_t kmin_visible) { return key ? key - kmin_visible + 1u : 0u; }
__device__ __forceinline__ uint32_t depth_key_restore(uint32_t adj, uint32_t kmin_visible) { return adj ? adj - 1u + kmin_visible : 0u; }

// K0: workgroup b finds splitter b = the sample of rank DSPL_PER b + DSPL_PER - 1 by a radix SELECT over the samples it holds in registers
// (most-significant-digit passes: histogram of the candidates' digit in LDS, scan, the digit whose run holds the rank) -- 255 independent
// workgroups on an otherwise idle chip.  The select runs on (sample - smallest sample) and only over the bytes the samples' range needs: depth keys
// share their top bytes, and a pass on a byte that every sample shares is 4096 atomic adds on ONE LDS word (the same select over all four bytes
// of the raw keys: 12 us; one workgroup sorting the samples with a bitonic network: 13; eight workgroups ranking them by counting: 53).
template <int SAMPLES> // 4096, or 1024 for scenes whose buckets are small against K3's capacity whatever their balance
__global__ void __launch_bounds__(256) depth_split_sample_kernel(int P, GeometryStateView g, DepthSplit ds)
{
    constexpr int DSPL_SAMPLES = SAMPLES, DSPL_PER = SAMPLES / NB;
    __shared__ uint32_t hist[NB], wtot[4], pick[2], rmin[4], rmax[4];
    constexpr int K = DSPL_SAMPLES / 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t *keys = (const uint32_t *)g.depth;
    const int nblocks = (P + SB - 1) / SB;
    for (int i = blockIdx.x * 256 + t; i < nblocks; i += gridDim.x * 256) g.blocksum[i] = 0ull; // K3 adds into them
    uint32_t smp[K];
#pragma unroll
    for (int k = 0; k < K; k++) smp[k] = keys[(int)(((unsigned long long)(t + 256 * k) * (unsigned long long)P) / DSPL_SAMPLES)];
    // culled triangles carry key 0: they stay 0, the visible ones become key - (smallest visible key) + 1 (monotone)
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < K; k++) { kmin = min(kmin, smp[k] ? smp[k] : 0xFFFFFFFFu); kmax = max(kmax, smp[k]); }
    for (int o = 32; o > 0; o >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o)); }
    if (lane == 0) { rmin[wave] = kmin; rmax[wave] = kmax; }
    __syncthreads();
    kmin = min(min(rmin[0], rmin[1]), min(rmin[2], rmin[3]));
    kmax = max(max(rmax[0], rmax[1]), max(rmax[2], rmax[3]));
    const uint32_t range = kmax ? kmax - kmin + 1u : 0u; // of the adjusted samples (no visible sample: all 0)
    const int top = range == 0u ? -8 : 8 * ((31 - __clz((int)range)) / 8); // shift of the highest byte in which two samples differ
#pragma unroll
    for (int k = 0; k < K; k++) smp[k] = depth_key_adjust(smp[k], kmin);
    uint32_t r = (uint32_t)(DSPL_PER * blockIdx.x + DSPL_PER - 1), prefix = 0u, mask = 0u;
#pragma unroll 1
    for (int shift = top; shift >= 0; shift -= 8)
    {
        hist[t] = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++)
            if ((smp[k] & mask) == prefix) atomicAdd(&hist[(smp[k] >> shift) & 0xFFu], 1u);
        __syncthreads();
        const uint32_t c = hist[t];
        const uint32_t inc = wave_inclusive_scan(c, lane);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        uint32_t excl = inc - c;
        for (int w = 0; w < wave; w++) excl += wtot[w];
        if (excl <= r && r < excl + c) { pick[0] = (uint32_t)t; pick[1] = r - excl; } // exactly one digit's run holds the rank
        __syncthreads();
        prefix |= pick[0] << shift;
        mask |= 0xFFu << shift;
        r = pick[1];
    }
    if (t == 0) ds.splitters[blockIdx.x] = depth_key_restore(prefix, kmin);
    if (blockIdx.x == 0 && t == 1) ds.splitters[NB - 1] = 0xFFFFFFFFu;
}

// number of splitters below the key: the first index whose splitter is >= key (sp[255] = ~0 stops every search)
__device__ __forceinline__ uint32_t depth_bucket_of(const uint32_t *sp, uint32_t key)
{
    uint32_t lo = 0u;
#pragma unroll
    for (int step = NB / 2; step > 0; step >>= 1)
        lo += (sp[lo + step - 1] < key) ? step : 0u;
    return lo;
}

template <int CH>
__global__ void __launch_bounds__(256) depth_split_hist_kernel(int64_t n, GeometryStateView g, RadixScratchView r, uint32_t *__restrict__ acc, DepthSplit ds)
{
    __shared__ uint32_t bins[NB], sp[NB];
    __shared__ unsigned long long csum[4];
    const int t = threadIdx.x, chunk = rs_chunk_of_block(r.chunks);
    if (chunk < 0) return;
    bins[t] = 0u;
    sp[t] = ds.splitters[t];
    __syncthreads();
    const uint32_t *keys = (const uint32_t *)g.depth;
    const int64_t base = (int64_t)chunk * CH;
    unsigned long long tsum = 0;
#pragma unroll
    for (int b = 0; b < CH / 256; b++)
    {
        const int64_t i = base + 256 * b + t;
        if (i < n)
        {
            const uint32_t d = depth_bucket_of(sp, keys[i]);
            ds.digit[i] = (uint8_t)d;
            atomicAdd(&bins[d], 1u);
            tsum += g.tiles_touched[i];
        }
    }
    for (int o = 32; o > 0; o >>= 1) tsum += __shfl_xor(tsum, o);
    if ((t & 63) == 0) csum[t >> 6] = tsum;
    __syncthreads();
    const uint32_t c = bins[t];
    r.table[(size_t)chunk * NB + t] = c;
    if (c) __hip_atomic_fetch_add(acc + (size_t)(chunk >> 6) * NB + t, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == 0) ds.chunk_sum[chunk] = csum[0] + csum[1] + csum[2] + csum[3];
}

// rs_scatter_body<IDENTITY_VALUES, CH, DIRECT> with the digit read from K1's byte array (and parked in LDS beside the pair, for the way out)
template <int CH>
__global__ void __launch_bounds__(256) depth_split_scatter_kernel(int64_t n, GeometryStateView g, RadixScratchView r, const uint32_t *__restrict__ acc, DepthSplit ds,
                                                                   unsigned long long *host_out)
{
    constexpr int KB = CH / 256;
    constexpr int SK = CH < 8 * NB ? 8 * NB : CH, SV = CH < 4 * NB ? 4 * NB : CH;
    __shared__ __attribute__((aligned(16))) uint32_t stage_k[SK], stage_v[SV];
    __shared__ uint8_t stage_d[CH];
    __shared__ uint32_t wcnt[4][NB];
    __shared__ int32_t gdelta[NB];
    __shared__ uint32_t wtot[4], gtot[4];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int chunk = rs_chunk_of_block(r.chunks);
    if (chunk < 0) return;
    const uint32_t *kin = (const uint32_t *)g.depth;
    const int64_t base = (int64_t)chunk * CH + (int64_t)wave * (CH / 4);
    const int64_t here = n - base;
    const int mine = here >= CH / 4 ? CH / 4 : (here > 0 ? (int)here : 0);
    uint32_t key[KB], dg[KB], rk[KB];
#pragma unroll
    for (int b = 0; b < KB; b++)
    {
        const int i = 64 * b + lane;
        key[b] = 0xFFFFFFFFu;
        dg[b] = 0u;
        if (i < mine) { key[b] = kin[base + i]; dg[b] = ds.digit[base + i]; }
    }
    uint4 p_within = make_uint4(0u, 0u, 0u, 0u), p_before = p_within, p_total = p_within;
    {
        const int slab = chunk >> 6, c0 = slab * 64;
        const uint4 *tab4 = (const uint4 *)r.table + (size_t)c0 * (NB / 4) + lane;
        const uint4 *acc4 = (const uint4 *)acc + lane;
#pragma unroll 4
        for (int k = 0; k < 16; k++)
        {
            const int c = wave + 4 * k;
            if (c0 + c < chunk)
            {
                const uint4 v = tab4[(size_t)c * (NB / 4)];
                p_within.x += v.x; p_within.y += v.y; p_within.z += v.z; p_within.w += v.w;
            }
        }
#pragma unroll 4
        for (int k = 0; k < TS_DIRECT_MAX_SLABS / 4; k++)
        {
            const int sl = wave + 4 * k;
            if (sl < r.slabs)
            {
                const uint4 v = acc4[(size_t)sl * (NB / 4)];
                p_total.x += v.x; p_total.y += v.y; p_total.z += v.z; p_total.w += v.w;
                if (sl < slab) { p_before.x += v.x; p_before.y += v.y; p_before.z += v.z; p_before.w += v.w; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NB / 64; k++) wcnt[wave][lane + 64 * k] = 0u;
    wave_lds_order();
    uint32_t *cnt = wcnt[wave];
#pragma unroll
    for (int b = 0; b < KB; b++)
    {
        const bool valid = 64 * b + lane < mine;
        const uint32_t d = dg[b];
        const unsigned long long vm = ballot64(valid);
        uint32_t mis_lo = ~(uint32_t)vm, mis_hi = ~(uint32_t)(vm >> 32);
#pragma unroll
        for (int bit = 0; bit < 8; bit++)
        {
            const unsigned long long bb = ballot64((d >> bit) & 1u);
            const uint32_t e = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit, 1);
            mis_lo |= (uint32_t)bb ^ e;
            mis_hi |= (uint32_t)(bb >> 32) ^ e;
        }
        const uint32_t m_lo = ~mis_lo, m_hi = ~mis_hi;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const uint32_t c = (uint32_t)(__popc(m_lo) + __popc(m_hi));
        uint32_t seen = 0;
        if (valid) seen = cnt[d];
        wave_lds_order();
        if (valid && rank == c - 1u) cnt[d] = seen + c;
        wave_lds_order();
        rk[b] = seen + rank;
    }
    *(uint4 *)(stage_k + wave * NB + 4 * lane) = p_within;
    *(uint4 *)(stage_k + (4 + wave) * NB + 4 * lane) = p_before;
    *(uint4 *)(stage_v + wave * NB + 4 * lane) = p_total;
    __syncthreads();
    {
        uint32_t d_within = 0u, d_before = 0u, d_total = 0u; // thread t = bucket t
#pragma unroll
        for (int w = 0; w < 4; w++)
        {
            d_within += stage_k[w * NB + t];
            d_before += stage_k[(4 + w) * NB + t];
            d_total += stage_v[w * NB + t];
        }
        const uint32_t c0 = wcnt[0][t], c1 = wcnt[1][t], c2 = wcnt[2][t], c3 = wcnt[3][t];
        const uint32_t tot = c0 + c1 + c2 + c3;
        const uint32_t inc = wave_inclusive_scan(tot, lane);
        if (lane == 63) wtot[wave] = inc;
        const uint32_t ginc = wave_inclusive_scan(d_total, lane);
        if (lane == 63) gtot[wave] = ginc;
        __syncthreads();
        uint32_t dbase = inc - tot, gbase = ginc - d_total;
        for (int w = 0; w < wave; w++) { dbase += wtot[w]; gbase += gtot[w]; }
        wcnt[0][t] = dbase; wcnt[1][t] = dbase + c0; wcnt[2][t] = dbase + c0 + c1; wcnt[3][t] = dbase + c0 + c1 + c2;
        gdelta[t] = (int32_t)(gbase + d_before + d_within - dbase);
        if (chunk == 0) ds.bucket_start[t] = gbase; // where bucket t begins in sk[0] / sv[0]
    }
    __syncthreads();
    const int64_t left = n - (int64_t)chunk * CH;
    const int count = left < CH ? (int)left : CH;
#pragma unroll
    for (int b = 0; b < KB; b++)
        if (64 * b + lane < mine)
        {
            const uint32_t p = wcnt[wave][dg[b]] + rk[b];
            stage_k[p] = key[b];
            stage_v[p] = (uint32_t)(base + 64 * b + lane);
            stage_d[p] = (uint8_t)dg[b];
        }
    __syncthreads();
    for (int p = t; p < count; p += 256)
    {
        const int64_t dst = (int64_t)gdelta[stage_d[p]] + p;
        g.sk[0][dst] = stage_k[p];
        g.sv[0][dst] = stage_v[p];
    }
    if (chunk == 0) // N = the chunks' sums of K1: to the device word the scan reads and to the pinned host word (publish_census of the LSD form)
    {
        __shared__ unsigned long long psum[4];
        unsigned long long sum = 0;
        for (int c = t; c < r.chunks; c += 256) sum += ds.chunk_sum[c];
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) psum[wave] = sum;
        __syncthreads();
        if (t == 0)
        {
            const unsigned long long N = psum[0] + psum[1] + psum[2] + psum[3];
            g.blocksum[((int)n + SB - 1) / SB] = N; // behind the block sums K3 adds up (= DepthCensus::n_out)
            *g.top_const = 0u;                      // the order is always in sk[1] / sv[1]
            if (host_out) __hip_atomic_store(host_out, N, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// One stable LSD pass of a 16-wave workgroup over the `mine` pairs each wave holds in registers (wave w: positions [w per, w per + mine) of the
// tile, 64 consecutive ones per step), on digit (key - kmin) >> shift: the ranking of depth_order_small_kernel.  Leaves, for every pair, its
// position among the tile's pairs in (digit, wave, step, lane) order in rk[], and the tile's digit counts in tile_cnt (thread d < 256: digit d).
template <int KBX, int W>
__device__ __forceinline__ void bucket_rank_pass(const uint32_t (&key)[KBX], uint32_t (&rk)[KBX], uint32_t kmin, int shift, int nbits, int per, int mine,
                                                 uint32_t (*wcnt)[NB], uint32_t *wtot, uint32_t &tile_cnt)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    uint32_t *cnt = wcnt[wave];
#pragma unroll
    for (int k = 0; k < NB / 64; k++) cnt[lane + 64 * k] = 0u;
    wave_lds_order();
#pragma unroll
    for (int b = 0; b < KBX; b++)
    {
        if (64 * b >= per) continue; // wave-uniform
        const bool valid = 64 * b + lane < mine;
        const uint32_t d = ((key[b] - kmin) >> shift) & 0xFFu;
        const unsigned long long vm = ballot64(valid);
        uint32_t mis_lo = ~(uint32_t)vm, mis_hi = ~(uint32_t)(vm >> 32);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) // unrolled over all eight bits (see rs_scatter_body); `nbits` is not needed
        {
            const unsigned long long bb = ballot64((d >> bit) & 1u);
            const uint32_t e = (uint32_t)__builtin_amdgcn_sbfe((int)d, bit, 1);
            mis_lo |= (uint32_t)bb ^ e;
            mis_hi |= (uint32_t)(bb >> 32) ^ e;
        }
        const uint32_t m_lo = ~mis_lo, m_hi = ~mis_hi;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
        const uint32_t c = (uint32_t)(__popc(m_lo) + __popc(m_hi));
        uint32_t seen = 0;
        if (valid) seen = cnt[d];
        wave_lds_order();
        if (valid && rank == c - 1u) cnt[d] = seen + c;
        wave_lds_order();
        rk[b] = seen + rank;
    }
    __syncthreads();
    uint32_t c[W], tot = 0u, inc = 0u;
    if (t < NB) // thread d = digit d
    {
#pragma unroll
        for (int w = 0; w < W; w++) { c[w] = wcnt[w][t]; tot += c[w]; }
        inc = wave_inclusive_scan(tot, lane);
        if (lane == 63) wtot[wave] = inc;
    }
    __syncthreads();
    if (t < NB)
    {
        uint32_t run = inc - tot;
        for (int w = 0; w < wave; w++) run += wtot[w];
#pragma unroll
        for (int w = 0; w < W; w++) { wcnt[w][t] = run; run += c[w]; }
    }
    tile_cnt = tot;
    __syncthreads();
#pragma unroll
    for (int b = 0; b < KBX; b++)
        if (64 * b + lane < mine) rk[b] += cnt[((key[b] - kmin) >> shift) & 0xFFu];
}

// What gather_blocksum_kernel does, for 64 consecutive depth-order positions (first one `pos0`, this lane's `pos`): the tile counts in depth order
// and their sums per scan block -- 64 consecutive positions meet at most two blocks; the sums were cleared by K0.
// `lsum` (fast path): the workgroup's scan-block sums in LDS, entry 0 = the block of the bucket's first position; they leave as ONE global atomic
// per (bucket, scan block) at the end (2 k atomics at 1 M triangles instead of 16 k, one or two per wave and step).  Null: straight to memory.
__device__ __forceinline__ void bucket_emit_tiles(const GeometryStateView &g, int pos0, int pos, bool in, uint32_t tt, int lane, uint32_t *lsum = nullptr,
                                                  int lblk0 = 0)
{
    if (in) g.tiles_sorted[pos] = tt;
    // 64 tile counts fit 32 bits together (their sum is at most N, and a forward with N >= 2^31 is refused on the host).  The positions grow with
    // the lane, so the lanes of the first scan block are a prefix of the wave: its sum is the inclusive scan at the last of them, the second block's
    // the rest -- one DPP scan (a 64-bit butterfly per block took 24 ds_bpermute per step).
    const int blk0 = pos0 / SB;
    const uint32_t inc = wave_inclusive_scan(in ? tt : 0u, lane);
    const int nfirst = min(64, (blk0 + 1) * SB - pos0); // lanes whose position lies in blk0 (>= 1)
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63), a = (uint32_t)__builtin_amdgcn_readlane((int)inc, nfirst - 1);
    if (lane == 0)
    {
        if (lsum)
        {
            if (a) atomicAdd(&lsum[blk0 - lblk0], a);
            if (total - a) atomicAdd(&lsum[blk0 - lblk0 + 1], total - a);
        }
        else
        {
            if (a) __hip_atomic_fetch_add((unsigned long long *)g.blocksum + blk0, (unsigned long long)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (total - a) __hip_atomic_fetch_add((unsigned long long *)g.blocksum + blk0 + 1, (unsigned long long)(total - a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static_assert(2 * DB_CAP * 4 + DB_WAVES * NB * 4 + 1024 <= 160 * 1024 - 8 * 1024, "depth_bucket_sort_kernel: static LDS must leave 8 KB of the 160 KB");
// W waves per workgroup: 16 (64 KB + 64 KB of staging, one workgroup per compute unit) for buckets of thousands of pairs, 4 for the buckets of a
// few hundred that scenes below DB_SMALL_BELOW triangles have (a barrier among four waves, four columns per digit in the scan)
template <int W>
__global__ void __launch_bounds__(64 * W) depth_bucket_sort_kernel(int P, GeometryStateView g, DepthSplit ds, int cap)
{
    constexpr int DB_WAVES = W, DB_CAP = 64 * W * DB_KB;
    __shared__ __attribute__((aligned(16))) uint32_t stage_k[DB_CAP], stage_v[DB_CAP];
    __shared__ uint32_t wcnt[DB_WAVES][NB];
    __shared__ uint32_t wtot[4], rmin[DB_WAVES], rmax[DB_WAVES], dstart[NB];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int bucket = blockIdx.x;
    const int s0 = (int)ds.bucket_start[bucket], e0 = bucket + 1 < NB ? (int)ds.bucket_start[bucket + 1] : P;
    const int n = e0 - s0;
    if (n <= 0) return;
    const bool fits = n <= cap;
    // the whole bucket in registers when it fits (wave w: `per` consecutive positions, 64 per step); the key range either way
    const int per = ((n + 64 * DB_WAVES - 1) / (64 * DB_WAVES)) * 64;
    const int base = wave * per;
    const int mine = !fits ? 0 : (n - base < per ? (n - base > 0 ? n - base : 0) : per);
    uint32_t key[DB_KB], val[DB_KB], rk[DB_KB];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u; // kmin: over the VISIBLE keys (depth_key_adjust)
    if (fits)
    {
        // branch-free loads (a lane past the end reads the bucket's first pair): all of a wave's steps are in flight together.  (With the loads under
        // `in ? load : constant` and the minimum taken in the same loop, every step waited for its predecessor: ten round trips in a row.)
#pragma unroll
        for (int b = 0; b < DB_KB; b++)
        {
            if (64 * b >= per) { key[b] = 0xFFFFFFFFu; val[b] = 0u; continue; } // wave-uniform
            const int i = 64 * b + lane;
            const int src = s0 + (i < mine ? base + i : 0);
            key[b] = g.sk[0][src];
            val[b] = g.sv[0][src];
        }
#pragma unroll
        for (int b = 0; b < DB_KB; b++)
        {
            const bool in = 64 * b + lane < mine;
            kmin = min(kmin, in && key[b] ? key[b] : 0xFFFFFFFFu);
            kmax = max(kmax, in ? key[b] : 0u);
        }
    }
    else
        for (int i = t; i < n; i += 64 * DB_WAVES)
        {
            const uint32_t k = g.sk[0][s0 + i];
            kmin = min(kmin, k ? k : 0xFFFFFFFFu);
            kmax = max(kmax, k);
        }
    for (int o = 32; o > 0; o >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o)); }
    if (lane == 0) { rmin[wave] = kmin; rmax[wave] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DB_WAVES; w++) { kmin = min(kmin, rmin[w]); kmax = max(kmax, rmax[w]); }
    // sorted on the adjusted keys (culled 0, visible key - kmin + 1): `range` = the largest of them
    const uint32_t range = kmax ? kmax - kmin + 1u : 0u;
    const int kbits = range == 0u ? 0 : 32 - __clz((int)range), passes = (kbits + 7) / 8; // the bits that vary inside the bucket
    if (fits)
    {
#pragma unroll
        for (int b = 0; b < DB_KB; b++) key[b] = 64 * b + lane < mine ? depth_key_adjust(key[b], kmin) : 0xFFFFFFFFu;
    }
    if (fits)
    {
        // after every pass the pairs change owners through LDS
#pragma nounroll
        for (int pass = 0; pass < passes; pass++)
        {
            uint32_t unused;
            bucket_rank_pass<DB_KB, W>(key, rk, 0u, 8 * pass, min(8, kbits - 8 * pass), per, mine, wcnt, wtot, unused);
#pragma unroll
            for (int b = 0; b < DB_KB; b++)
                if (64 * b + lane < mine) { stage_k[rk[b]] = key[b]; stage_v[rk[b]] = val[b]; }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < DB_KB; b++)
                if (64 * b < per) { key[b] = stage_k[base + 64 * b + lane]; val[b] = stage_v[base + 64 * b + lane]; }
            __syncthreads();
        }
        // the tile counts of all steps are requested before anything else happens: one gather latency per wave, not one per step
#pragma unroll
        for (int b = 0; b < DB_KB; b++)
            rk[b] = 64 * b >= per ? 0u : g.tiles_touched[64 * b + lane < mine ? val[b] : 0u];
#pragma unroll
        for (int b = 0; b < DB_KB; b++)
        {
            if (64 * b >= per) continue;
            const int pos = s0 + base + 64 * b + lane;
            if (64 * b + lane < mine) { g.sk[1][pos] = depth_key_restore(key[b], kmin); g.sv[1][pos] = val[b]; }
        }
        // (a bucket of n <= DB_CAP pairs meets at most DB_CAP / SB + 2 scan blocks; a block's sum stays below N < 2^31)
        constexpr int LB = DB_CAP / SB + 2;
        uint32_t *lsum = dstart; // free on this path (NB >= LB words)
        static_assert(LB <= NB, "the scan-block sums borrow dstart");
        if (t < LB) lsum[t] = 0u;
        __syncthreads();
        const int lblk0 = s0 / SB;
#pragma unroll
        for (int b = 0; b < DB_KB; b++)
        {
            if (64 * b >= per) continue;
            bucket_emit_tiles(g, s0 + base + 64 * b, s0 + base + 64 * b + lane, 64 * b + lane < mine, 64 * b + lane < mine ? rk[b] : 0u, lane, lsum, lblk0);
        }
        __syncthreads();
        if (t < LB && lsum[t])
            __hip_atomic_fetch_add((unsigned long long *)g.blocksum + lblk0 + t, (unsigned long long)lsum[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    else
    {
        // larger than the registers: the same passes through global memory, tile after tile, between (sk[0], sv[0]) and (sk[1], sv[1]) of this
        // bucket's range (nobody else touches it); an odd number of passes ends in [1], an even one -- and a bucket of equal keys -- is copied
        const int tile = (cap & ~(64 * DB_WAVES - 1)) ? (cap & ~(64 * DB_WAVES - 1)) : 64 * DB_WAVES; // pairs per tile: whole 64-pair steps per wave
        int src = 0;
        for (int pass = 0; pass < passes; pass++, src ^= 1)
        {
            const uint32_t *kin = g.sk[src] + s0, *vin = g.sv[src] + s0;
            uint32_t *kout = g.sk[src ^ 1] + s0, *vout = g.sv[src ^ 1] + s0;
            if (t < NB) dstart[t] = 0u;
            __syncthreads();
            for (int i = t; i < n; i += 64 * DB_WAVES) atomicAdd(&dstart[(depth_key_adjust(peer_load(kin + i), kmin) >> (8 * pass)) & 0xFFu], 1u);
            __syncthreads();
            {
                uint32_t tot = 0u, inc = 0u;
                if (t < NB) { tot = dstart[t]; inc = wave_inclusive_scan(tot, lane); if (lane == 63) wtot[wave] = inc; }
                __syncthreads();
                if (t < NB)
                {
                    uint32_t run = inc - tot;
                    for (int w = 0; w < wave; w++) run += wtot[w];
                    dstart[t] = run; // where digit t begins in the bucket
                }
                __syncthreads();
            }
            for (int t0 = 0; t0 < n; t0 += tile)
            {
                const int nt = n - t0 < tile ? n - t0 : tile;
                const int tper = ((nt + 64 * DB_WAVES - 1) / (64 * DB_WAVES)) * 64;
                const int tbase = wave * tper;
                const int tmine = nt - tbase < tper ? (nt - tbase > 0 ? nt - tbase : 0) : tper;
#pragma unroll
                for (int b = 0; b < DB_KB; b++)
                {
                    const int i = 64 * b + lane;
                    const bool in = i < tmine;
                    key[b] = in ? depth_key_adjust(peer_load(kin + t0 + tbase + i), kmin) : 0xFFFFFFFFu; // past this compute unit's L1: an earlier pass of this workgroup wrote them
                    val[b] = in ? peer_load(vin + t0 + tbase + i) : 0u;
                }
                uint32_t tile_cnt;
                bucket_rank_pass<DB_KB, W>(key, rk, 0u, 8 * pass, min(8, kbits - 8 * pass), tper, tmine, wcnt, wtot, tile_cnt);
                // rk = position inside the tile's (digit, order) arrangement; the digit's pairs of this tile go behind the earlier tiles' ones:
                // global position = dstart[d] + (rk - start of d inside the tile) = dstart[d] + rk - wcnt[0][d]
#pragma unroll
                for (int b = 0; b < DB_KB; b++)
                    if (64 * b + lane < tmine)
                    {
                        const uint32_t d = (key[b] >> (8 * pass)) & 0xFFu;
                        const uint32_t p = dstart[d] + rk[b] - wcnt[0][d];
                        peer_store(kout + p, depth_key_restore(key[b], kmin));
                        peer_store(vout + p, val[b]);
                    }
                __syncthreads();
                if (t < NB) dstart[t] += tile_cnt;
                __syncthreads();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the next pass of this workgroup reads what other waves of it wrote to global memory
            __syncthreads();
        }
        if (src == 0)
            for (int i = t; i < n; i += 64 * DB_WAVES) { peer_store(g.sk[1] + s0 + i, peer_load(g.sk[0] + s0 + i)); peer_store(g.sv[1] + s0 + i, peer_load(g.sv[0] + s0 + i)); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i0 = 64 * wave; i0 < n; i0 += 64 * DB_WAVES)
        {
            const bool in = i0 + lane < n;
            bucket_emit_tiles(g, s0 + i0, s0 + i0 + lane, in, in ? g.tiles_touched[peer_load(g.sv[1] + s0 + i0 + lane)] : 0u, lane);
        }
    }
}
} // namespace

// Step 1: (depth bits, id) -> sorted ids.  Depth keys are view-space z of visible triangles (> 0, so the unsigned bit
// pattern is monotone) and 0 for culled ones, which emit nothing wherever they land.  `begin` = the launches that produce N (to `host_out`
// as well when given) and the key-bit census: the first histogram, and on the ticket-free path the first scatter too; `finish` = the other
// launches (the 4th pass returns at once when the census found the top byte constant: then sk[0] / sv[0] hold the order).
// The one-launch form (depth_order_small_kernel) -- never under the lab library's switches, which exist to run the multi-launch forms on small scenes.
#ifndef TS_DEPTH_SMALL_USE_VALUE // up to how many triangles the one-launch form is USED (<= its capacity TS_DEPTH_SMALL_MAX)
#define TS_DEPTH_SMALL_USE_VALUE 9216 // measured against the split form (variant builds, 256 x 256, graph replay): 3 k 17 vs 30 us, 5 k 22 vs 30, 10 k 32 vs 30 (step 0.133 -> 0.128 ms)
#endif
constexpr int TS_DEPTH_SMALL_USE = TS_DEPTH_SMALL_USE_VALUE;
static_assert(TS_DEPTH_SMALL_USE <= TS_DEPTH_SMALL_MAX && TS_DEPTH_SMALL_USE >= 2048, "the split form's scratch needs 1024 + 2 chunks words of `offsets`");
static bool depth_small_ok(int32_t P) { return P <= TS_DEPTH_SMALL_USE && !g_force_tickets && !g_force_pass4; }
// The sampled-splitter form: every size between the one-launch form and TS_DEPTH_SPLIT_MAX whose scratch takes the ticket-free passes; never under the
// lab library's switches that ask for the LSD forms.
static bool depth_split_ok(int32_t P, const RadixScratchView &r)
{
    return P > TS_DEPTH_SMALL_USE && P <= (g_depth_split_mode == 2 ? TS_DEPTH_SPLIT_HARD_MAX : TS_DEPTH_SPLIT_MAX) && g_depth_split_mode != 1 && !g_force_tickets &&
           !g_force_pass4 && radix_direct_ok(r);
}
void ts_sort_by_depth_begin(const GeometryStateView &g, int32_t P, unsigned long long *host_out, hipStream_t s)
{
    if (P <= 0) return;
    if (depth_small_ok(P))
    {
        hipLaunchKernelGGL(depth_order_small_kernel, dim3(1), dim3(64 * DS_WAVES), 0, s, P, g, host_out);
        return;
    }
    if (depth_split_ok(P, g.rs))
    {
        const DepthSplit ds = depth_split_of(g);
        const dim3 grid((unsigned)g.rs.chunks);
        if (P < DSPL_SMALL_BELOW) hipLaunchKernelGGL(depth_split_sample_kernel<DSPL_SAMPLES_SMALL>, dim3(NB - 1), dim3(256), 0, s, P, g, ds);
        else hipLaunchKernelGGL(depth_split_sample_kernel<DSPL_SAMPLES>, dim3(NB - 1), dim3(256), 0, s, P, g, ds);
        TS_WITH_CHUNK(g.rs.chunk,
                      hipLaunchKernelGGL((depth_split_hist_kernel<CH>), grid, dim3(256), 0, s, (int64_t)P, g, g.rs, g.rs.slabacc[0], ds);
                      hipLaunchKernelGGL((depth_split_scatter_kernel<CH>), grid, dim3(256), 0, s, (int64_t)P, g, g.rs, (const uint32_t *)g.rs.slabacc[0], ds, host_out))
        return; // N is out; the buckets' sorts belong to `finish`
    }
    DepthCensus c;
    c.tiles_touched = g.tiles_touched;
    c.chunk_sum = (unsigned long long *)g.blocksum; // scratch: rewritten by gather_blocksum_kernel (chunks <= ceil(P / 1024))
    c.chunk_or = g.tiles_sorted;                    // scratch: rewritten by gather_blocksum_kernel / scan_emit_kernel
    c.chunk_and = g.offsets;
    c.n_out = (unsigned long long *)g.blocksum + (P + SB - 1) / SB;
    c.host_out = host_out;
    c.top_const = g.top_const;
    c.force_varying = g_force_pass4 ? 0xFF000000u : 0u;
    if (!radix_direct_ok(g.rs))
    {
        radix_hist((const uint32_t *)g.depth, P, nullptr, 0, 8, g.rs, s, &c);
        return;
    }
    // ticket-free: the histogram (with the census) adds into slabacc[0], which the step's first launch cleared (clear_tickets); the first
    // scatter belongs to `begin` because its block 0 publishes the census
    const dim3 grid((unsigned)g.rs.chunks);
    const uint32_t *keys = (const uint32_t *)g.depth;
    const unsigned long long *no_count = nullptr;
    const uint32_t *no_vals = nullptr, *no_skip = nullptr;
    TS_WITH_CHUNK(g.rs.chunk,
                  hipLaunchKernelGGL((rs_hist_census_direct_kernel<CH>), grid, dim3(256), 0, s, keys, (int64_t)P, 0xFFu, g.rs, g.rs.slabacc[0], c);
                  hipLaunchKernelGGL((rs_scatter_kernel<true, CH, true, true>), grid, dim3(256), 0, s, keys, no_vals, g.sk[0], g.sv[0], (int64_t)P, no_count, 0, 8,
                                     g.rs, no_skip, (const uint32_t *)g.rs.slabacc[0], g.rs.slabacc[1], c))
}
void ts_sort_by_depth_finish(const GeometryStateView &g, int32_t P, hipStream_t s)
{
    if (P <= 0 || depth_small_ok(P)) return;
    if (depth_split_ok(P, g.rs))
    {
        if (P < DB_SMALL_BELOW)
            hipLaunchKernelGGL(depth_bucket_sort_kernel<DB_SMALL_WAVES>, dim3(NB), dim3(64 * DB_SMALL_WAVES), 0, s, P, g, depth_split_of(g),
                               min(g_depth_bucket_cap, 64 * DB_SMALL_WAVES * DB_KB));
        else hipLaunchKernelGGL(depth_bucket_sort_kernel<DB_WAVES>, dim3(NB), dim3(64 * DB_WAVES), 0, s, P, g, depth_split_of(g), g_depth_bucket_cap);
        return;
    }
    if (!radix_direct_ok(g.rs))
    {
        radix_scatter((const uint32_t *)g.depth, nullptr, g.sk[0], g.sv[0], P, nullptr, 0, 8, g.rs, s);
        radix_pass(g.sk[0], g.sv[0], g.sk[1], g.sv[1], P, nullptr, 8, 8, g.rs, s);
        radix_pass(g.sk[1], g.sv[1], g.sk[0], g.sv[0], P, nullptr, 16, 8, g.rs, s);
        radix_pass(g.sk[0], g.sv[0], g.sk[1], g.sv[1], P, nullptr, 24, 8, g.rs, s, g.top_const);
        return;
    }
    radix_pass_direct(g.sk[0], g.sv[0], g.sk[1], g.sv[1], P, nullptr, 8, 8, g.rs, 1, s);
    radix_pass_direct(g.sk[1], g.sv[1], g.sk[0], g.sv[0], P, nullptr, 16, 8, g.rs, 0, s);
    radix_pass_direct(g.sk[0], g.sv[0], g.sk[1], g.sv[1], P, nullptr, 24, 8, g.rs, 1, s, g.top_const);
}

// Step 2: tiles_sorted = tiles_touched[perm], raw block sums (+ their groups' sums above 2048 blocks, or always under the lab library's
// ts2d_lab_force_ticket_passes, so that the suite runs the two-level form on small scenes); blocksum[nblocks] = N comes from the census.
static bool scan_two_level(int32_t P) { return g_force_tickets || (P + SB - 1) / SB > 2048; }
void ts_scan_offsets(const GeometryStateView &g, int32_t P, hipStream_t s)
{
    if (P <= 0 || depth_small_ok(P) || depth_split_ok(P, g.rs)) return; // depth_order_small_kernel / depth_bucket_sort_kernel left tiles_sorted and the block sums behind
    hipLaunchKernelGGL(gather_blocksum_kernel, dim3((unsigned)((P + SB - 1) / SB)), dim3(256), 0, s, P, g, scan_two_level(P));
}

void ts_launch_emit_keys(int P, int grid_x, int ntiles, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                         float *contrib_sum, float *contrib_max, int64_t capacity, int32_t *status, const QuadMaskArgs &qmask, hipStream_t s)
{
    if (P <= 0) return;
    hipLaunchKernelGGL(scan_emit_kernel, dim3((unsigned)(((P + SB - 1) / SB) * 4)), dim3(256), 0, s, P, grid_x, ntiles, g, b, im.ranges, contrib_sum,
                       contrib_max, (long long)capacity, status, scan_two_level(P), qmask);
}
const unsigned long long *ts_instance_count_dev(const GeometryStateView &g, int P) { return (const unsigned long long *)(g.blocksum + (P + SB - 1) / SB); }

// Step 4: stable sort of the instances by tile id: ceil(bits / 8) passes, ping-pong from (k[0], v[0]).
void ts_sort_pairs(const BinningStateView &b, int64_t N, const unsigned long long *n_dev, int ntiles, hipStream_t s)
{
    if (N <= 0) return;
    const int bits = ts_tile_bits(ntiles);
    // the key bits are shared out evenly over the passes (1080p: 13 bits = 7 + 6, not 8 + 5): fewer digits in the first pass mean longer
    // runs per digit in a chunk (32 pairs instead of 16), i.e. better coalesced stores, and one ballot less per ranking step
    const int per = (bits + b.passes - 1) / b.passes;
    int src = 0;
    for (int p = 0; p < b.passes; p++)
    {
        const int shift = per * p, nbits = min(per, bits - shift);
        // ticket-free passes when the sort is small enough; scan_emit_kernel cleared slabacc[0] for the first one
        if (radix_direct_ok(b.rs)) radix_pass_direct(b.k[src], b.v[src], b.k[src ^ 1], b.v[src ^ 1], N, n_dev, shift, nbits, b.rs, p & 1, s);
        else radix_pass(b.k[src], b.v[src], b.k[src ^ 1], b.v[src ^ 1], N, n_dev, shift, nbits, b.rs, s);
        src ^= 1;
    }
}

void ts_launch_tile_ranges(int64_t N, const unsigned long long *n_dev, const BinningStateView &b, const ImageStateView &im, hipStream_t s)
{
    if (N <= 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((N + 1023) / 1024)), dim3(256), 0, s, N, n_dev, b.tile, im.ranges);
}

// ---- the same sort for other callers (knn.hip: 30-bit Morton codes) --------------------------------------------------------
static int generic_chunk(size_t n) { return ts_instance_chunk(n); } // few keys: more, shorter workgroups
size_t ts_radix_scratch_bytes(size_t n)
{
    RadixScratchView r{};
    char *p = nullptr;
    ts_carve_radix(p, n, r, generic_chunk(n));
    return (size_t)p + TS_ALIGN;
}
// Stable LSD sort of (key, value) pairs by key bits [0, end_bit).  k[0] / v[0] hold the input, k[1] / v[1] are the ping-pong partners;
// returns which pair holds the result (passes & 1).
int ts_radix_sort_pairs(uint32_t *const k[2], uint32_t *const v[2], size_t n, int end_bit, void *scratch, hipStream_t s, bool force_tickets)
{
    if (n == 0) return 0;
    RadixScratchView r{};
    char *p = (char *)ts_align_up((size_t)scratch);
    ts_carve_radix(p, n, r, generic_chunk(n));
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((r.slabs + TS_RS_TICKET_EXTRA + 255) / 256)), dim3(256), 0, s, r.tickets, (size_t)(r.slabs + TS_RS_TICKET_EXTRA));
    const bool direct = !force_tickets && radix_direct_ok(r);
    if (direct) hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)r.slabs), dim3(256), 0, s, r.slabacc[0], (size_t)(r.slabs * NB));
    const int passes = (end_bit + 7) / 8;
    int src = 0;
    for (int ps = 0; ps < passes; ps++)
    {
        if (direct) radix_pass_direct(k[src], v[src], k[src ^ 1], v[src ^ 1], (int64_t)n, nullptr, 8 * ps, min(8, end_bit - 8 * ps), r, ps & 1, s);
        else radix_pass(k[src], v[src], k[src ^ 1], v[src ^ 1], (int64_t)n, nullptr, 8 * ps, min(8, end_bit - 8 * ps), r, s);
        src ^= 1;
    }
    return src;
}

void ts_force_ticket_passes(bool on) { g_force_tickets = on; }
namespace
{
__global__ void __launch_bounds__(256) zero_words4_kernel(uint4 *p, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
} // namespace
void ts_launch_zero_words(uint32_t *p, size_t n, hipStream_t s) // n 32-bit words; 16-byte stores where the range allows
{
    if (n == 0) return;
    if (n % 4 == 0 && ((size_t)p & 15) == 0) hipLaunchKernelGGL(zero_words4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (uint4 *)p, n / 4);
    else hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}
void ts_force_depth_pass4(bool on) { g_force_pass4 = on; }
void ts_lab_depth_split(int mode, int bucket_cap)
{
    g_depth_split_mode = mode;
    g_depth_bucket_cap = bucket_cap > 0 && bucket_cap < DB_CAP ? bucket_cap : DB_CAP;
}
