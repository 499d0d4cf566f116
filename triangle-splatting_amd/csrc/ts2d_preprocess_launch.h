// ts2d_preprocess_launch.h -- kernel wrappers and launch policy shared by the 2D and 3D per-triangle kernels.
//
// A variant supplies a struct with two device functions working on ONE triangle whose vertex / SH rows and gradient rows
// may live in global memory or in LDS:
//   Body::fwd(a, radii, g, idx, vp, shp)
//   Body::bwd(a, radii, g, grad_rec, idx, vp, shp, ov, osh, dL_dcenter2D, dL_dfeature, dL_dopacity)
// The wrappers decide how rows travel (ts2d_stage.h): direct (any alignment, any M) or staged through LDS by single-wave
// workgroups (coalesced dwordx4 traffic).
#pragma once
#include "ts2d_common.h"
#include "ts2d_stage.h"
#include "ts2d_math.h"
#include "ts2d_sh.h"

#ifndef TS_PRE_OUT_REGS
#define TS_PRE_OUT_REGS 0
#endif
namespace ts
{
// What a launch of the per-triangle forward computes (api.hip: forward_bin_impl).  The ordering chain -- depth sort, scan, emission, tile sort --
// needs the integer state and the geometric part of the render record; the SH colour (192 of the 327 bytes per triangle the kernel moves at SH
// degree 3) is first read by the blend kernel.  PRE_NOCOLOUR runs on the caller's stream and leaves r g b / the clamp flags 0;
// preprocess_colour_kernel fills them in on the library's side stream BESIDE the ordering chain (a chain of latency-bound launches that leaves the
// HBM mostly idle), throttled to a few hundred resident waves so that it does not stretch that chain's memory round trips.  PRE_ALL = one launch
// (small scenes, feature mode, short SH rows, unaligned inputs).
enum { PRE_ALL = 0, PRE_NOCOLOUR = 1 };

// The binning kernels' "last block finishes" tickets (binning.hip) are zeroed by the first launch of the step.  Every grid has at
// least 64 threads and slabs + TS_RS_TICKET_EXTRA <= max(P, 64), so the threads beyond P of a tiny scene take part.  The slab totals of the
// depth sort's first (ticket-free) histogram and the group sums of the scan are cleared here as well.
__device__ __forceinline__ void clear_tickets(const GeometryStateView &g, int idx, int P)
{
    if (idx < g.rs.slabs + TS_RS_TICKET_EXTRA) g.rs.tickets[idx] = 0u;
    const int nthreads = (int)(gridDim.x * blockDim.x);
    for (int k = idx; k < g.rs.slabs * TS_RS_BINS; k += nthreads) g.rs.slabacc[0][k] = 0u;
    const int groups = (((P + 1023) >> 10) + 63) >> 6; // the scan's second level (binning.hip): one sum per 64 scan blocks of 1024 triangles
    for (int k = idx; k < groups + 1; k += nthreads) g.supersum[k] = 0ull;
}

template <class Body>
__global__ void __launch_bounds__(256) preprocess_fwd_direct_kernel(PreprocessArgs a, int32_t *__restrict__ radii, GeometryStateView g)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    clear_tickets(g, idx, a.P);
    if (idx >= a.P) return;
    Body::template fwd<PRE_ALL>(a, radii, g, idx, a.vertex + 9 * (size_t)idx, a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr, g.rec + 4 * (size_t)idx);
}

// vertex rows always staged; SH rows (SHROW = 3 M > 0 floats) travel either through LDS as well (SH_REGS = false) or straight into
// the lane's registers with SHROW / 4 dwordx4 loads (SH_REGS = true).  Both read the rows at the same rate in isolation
// (tools/sh_stage_bench.hip: 5.5-5.7 TB/s), but 12.5 KB of LDS per single-wave workgroup held the kernel at 2 waves per SIMD with
// 73 % of the wave cycles spent waiting (profiles/r02_notes.md); without it the registers are the only limit.
#ifndef TS_PRE_FWD_WAVES // register budget of the staged forward kernel in waves per SIMD (0: the compiler's choice, 105 registers = 4 waves at SH degree 3:
                         // 72.3 us at 1 M triangles; round 6, variant builds alternating on one box: 5 waves / 96 registers 76.5 us, 6 waves / 80 registers 139 us)
#define TS_PRE_FWD_WAVES 0
#endif
template <class Body, int SHROW, bool SH_REGS, int MODE>
#if TS_PRE_FWD_WAVES > 0
__global__ void __launch_bounds__(64, TS_PRE_FWD_WAVES) preprocess_fwd_staged_kernel(
#else
__global__ void __launch_bounds__(64) preprocess_fwd_staged_kernel(
#endif
    PreprocessArgs a, int32_t *__restrict__ radii, GeometryStateView g)
{
    __shared__ float s_v[64 * 9];
    __shared__ float s_sh[(SHROW > 0 && !SH_REGS) ? 64 * (SHROW + 1) : 1];
    const int lane = threadIdx.x, row0 = blockIdx.x * 64, idx = row0 + lane;
    float shr[(SHROW > 0 && SH_REGS) ? SHROW : 4];
    if (SHROW > 0 && SH_REGS && idx < a.P)
    {
        const float4 *rowp = (const float4 *)(a.shs + (size_t)idx * SHROW);
#pragma unroll
        for (int c = 0; c < SHROW / 4; c++) *(float4 *)(shr + 4 * c) = rowp[c];
    }
    stage_rows_in<9, 9>(s_v, a.vertex, row0, a.P, lane);
    if (SHROW > 0 && !SH_REGS) stage_rows_in<SHROW, SHROW + 1>(s_sh, a.shs, row0, a.P, lane);
    __syncthreads();
    clear_tickets(g, idx, a.P);
    // the 64 render records of the workgroup are one contiguous 4 KB block: each lane parks its record in LDS (row stride 80 bytes:
    // conflict-free 128-bit accesses) and the block leaves with coalesced dwordx4 stores instead of four stores at a 64-byte lane stride
    __shared__ float4 s_rec[64 * 5];
    if (idx < a.P)
    {
        const float *shp = SHROW > 0 ? (SH_REGS ? shr : s_sh + lane * (SHROW + 1)) : (a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr);
        Body::template fwd<MODE>(a, radii, g, idx, s_v + lane * 9, shp, s_rec + lane * 5);
    }
    __syncthreads();
    float4 *out = g.rec + 4 * (size_t)row0;
#pragma unroll
    for (int it = 0; it < 4; it++)
    {
        const int i = it * 64 + lane;
        if (row0 + (i >> 2) < a.P) out[i] = s_rec[(i >> 2) * 5 + (i & 3)];
    }
}

template <class Body>
__global__ void __launch_bounds__(256) preprocess_bwd_direct_kernel(PreprocessArgs a, const int32_t *__restrict__ radii,
                                                                     GeometryStateView g, const float *__restrict__ grad_rec,
                                                                     float *__restrict__ dL_dvertex, float *__restrict__ dL_dcenter2D,
                                                                     float *__restrict__ dL_dshs, float *__restrict__ dL_dfeature,
                                                                     float *__restrict__ dL_dopacity)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.P) return;
    Body::bwd(a, radii, g, grad_rec, idx, a.vertex + 9 * (size_t)idx, a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr,
              dL_dvertex + 9 * (size_t)idx, dL_dshs ? dL_dshs + (size_t)idx * a.M * 3 : nullptr, dL_dcenter2D, dL_dfeature,
              dL_dopacity);
}

// SHROW = 3 M > 0: LDS rows carry the SH coefficients in (SH_IN) and / or the dL_dshs rows out (WRITE_SH); the vertex rows
// carry the vertices in and dL_dvertex out.  A lane only ever touches its own rows between the two cooperative phases.
template <class Body, int SHROW, bool SH_IN, bool WRITE_SH, bool SH_REGS>
__global__ void __launch_bounds__(64) preprocess_bwd_staged_kernel(PreprocessArgs a, const int32_t *__restrict__ radii,
                                                                    GeometryStateView g, const float *__restrict__ grad_rec,
                                                                    float *__restrict__ dL_dvertex, float *__restrict__ dL_dcenter2D,
                                                                    float *__restrict__ dL_dshs, float *__restrict__ dL_dfeature,
                                                                    float *__restrict__ dL_dopacity)
{
    __shared__ float s_v[64 * 9];
    constexpr bool OUT_REGS = SH_REGS && TS_PRE_OUT_REGS; // gradient rows leave from registers (strided dwordx4 stores) or through LDS (coalesced)
    __shared__ float s_sh[(SHROW > 0 && (!SH_REGS || (WRITE_SH && !OUT_REGS))) ? 64 * (SHROW + 1) : 1];
    const int lane = threadIdx.x, row0 = blockIdx.x * 64, idx = row0 + lane;
    // SH_REGS (rows of whole 16-byte pieces): the coefficient row comes straight into registers and the gradient row leaves from
    // registers, SHROW / 4 dwordx4 each -- see preprocess_fwd_staged_kernel
    float shr[(SHROW > 0 && SH_REGS && SH_IN) ? SHROW : 4], osr[(SHROW > 0 && OUT_REGS && WRITE_SH) ? SHROW : 4];
    if (SHROW > 0 && SH_REGS && SH_IN && idx < a.P)
    {
        const float4 *rowp = (const float4 *)(a.shs + (size_t)idx * SHROW);
#pragma unroll
        for (int c = 0; c < SHROW / 4; c++) *(float4 *)(shr + 4 * c) = rowp[c];
    }
    stage_rows_in<9, 9>(s_v, a.vertex, row0, a.P, lane);
    if (SHROW > 0 && SH_IN && !SH_REGS) stage_rows_in<SHROW, SHROW + 1>(s_sh, a.shs, row0, a.P, lane);
    __syncthreads();
    if (idx < a.P)
    {
        const float *vp = s_v + lane * 9;
        const float *shp = (SHROW > 0 && SH_IN) ? (SH_REGS ? shr : s_sh + lane * (SHROW + 1)) : (a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr);
        if (SHROW > 0 && OUT_REGS && WRITE_SH)
        {
            // the gradient row is expanded here, from the clamp-masked colour gradient the per-triangle function hands back: the
            // row then never has its address taken inside that function and stays in registers.  The vertex row is consumed
            // first (dL_dvertex overwrites it in place).
            const f3 center = divf(add(add(f3{vp[0], vp[1], vp[2]}, f3{vp[3], vp[4], vp[5]}), f3{vp[6], vp[7], vp[8]}), 3.0f);
            const f3 masked = Body::bwd(a, radii, g, grad_rec, idx, vp, shp, s_v + lane * 9, (float *)nullptr, dL_dcenter2D, dL_dfeature, dL_dopacity);
#pragma unroll
            for (int k = 0; k < SHROW; k++) osr[k] = 0.0f;
            if (radii[idx] > 0) sh_grad_store(a.D, a.M, center, f3{a.campos[0], a.campos[1], a.campos[2]}, masked, osr);
            float4 *rowo = (float4 *)(dL_dshs + (size_t)idx * SHROW);
#pragma unroll
            for (int c = 0; c < SHROW / 4; c++) rowo[c] = *(const float4 *)(osr + 4 * c);
        }
        else
        {
            float *row = (SHROW > 0 && !OUT_REGS) ? s_sh + lane * (SHROW + 1) : nullptr;
            Body::bwd(a, radii, g, grad_rec, idx, vp, shp, s_v + lane * 9, WRITE_SH ? row : nullptr, dL_dcenter2D, dL_dfeature, dL_dopacity);
        }
    }
    __syncthreads();
    stage_rows_out<9, 9>(s_v, dL_dvertex, row0, a.P, lane);
    if (SHROW > 0 && WRITE_SH && !OUT_REGS) stage_rows_out<SHROW, SHROW + 1>(s_sh, dL_dshs, row0, a.P, lane);
}

// Staging policy: vertex rows whenever the pointers are 16-byte aligned; SH rows in when at least half of each row is
// active (otherwise the direct strided read of the active prefix moves fewer bytes); dL_dshs rows out always (every
// element is written).  M outside {1, 4, 9, 16} never stages SH rows.
static inline int staged_shrow(const PreprocessArgs &a) { return (a.use_shs && (a.M == 1 || a.M == 4 || a.M == 9 || a.M == 16)) ? 3 * a.M : 0; }

// The SH colour of the triangles that survived the culls, written into their render records behind a PRE_NOCOLOUR launch: floats REC_OFF..+2 of
// the record (2D: 7, 3D: 13) and the clamp flags.  A persistent grid of single-wave workgroups (grid = the throttle): wave w takes the 64-triangle
// blocks w, w + grid, ...; a lane whose triangle was culled (no tiles) loads nothing -- a view that sees a fraction of the scene reads that fraction
// of the SH rows.  Same expressions as the single launch (sh_to_rgb on the world-space centroid, forward.cu:165-171, 51-58): same bits.
template <int SHROW, int REC_OFF>
__global__ void __launch_bounds__(64) preprocess_colour_kernel(PreprocessArgs a, GeometryStateView g)
{
    static_assert(SHROW % 4 == 0, "rows of whole 16-byte pieces");
    const int lane = threadIdx.x;
    const f3 cp = {a.campos[0], a.campos[1], a.campos[2]};
    for (int row0 = blockIdx.x * 64; row0 < a.P; row0 += gridDim.x * 64)
    {
        const int idx = row0 + lane;
        if (idx >= a.P || g.tiles_touched[idx] == 0u) continue;
        float shr[SHROW];
        const float4 *rowp = (const float4 *)(a.shs + (size_t)idx * SHROW);
#pragma unroll
        for (int c = 0; c < SHROW / 4; c++) *(float4 *)(shr + 4 * c) = rowp[c];
        const float *vp = a.vertex + 9 * (size_t)idx;
        const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]};
        const f3 center = divf(add(add(v1, v2), v3), 3.0f);
        f3 rgb = sh_to_rgb(a.D, shr, center, cp);
        g.clamped[idx] = (uint8_t)((rgb.x < 0 ? 1 : 0) | (rgb.y < 0 ? 2 : 0) | (rgb.z < 0 ? 4 : 0));
        float *rec = (float *)(g.rec + 4 * (size_t)idx) + REC_OFF;
        rec[0] = fmaxf(rgb.x, 0.0f); rec[1] = fmaxf(rgb.y, 0.0f); rec[2] = fmaxf(rgb.z, 0.0f);
    }
}

// Can the SH colour of these arguments be split off (PRE_NOCOLOUR + preprocess_colour_kernel)?  SH rows of whole 16-byte pieces that are worth a
// launch of their own (M = 4 or 16), aligned inputs (the staged kernels).
static inline bool preprocess_fwd_splittable(const PreprocessArgs &a)
{
    return a.P > 0 && a.use_shs && (a.M == 4 || a.M == 16) && aligned16(a.vertex) && aligned16(a.shs);
}

template <int REC_OFF>
void launch_preprocess_colour(const PreprocessArgs &a, const GeometryStateView &g, int blocks, hipStream_t s)
{
    const int nb = (a.P + 63) / 64;
    const dim3 grid(blocks < nb ? blocks : nb), block(64);
    if (a.M == 16) hipLaunchKernelGGL((preprocess_colour_kernel<48, REC_OFF>), grid, block, 0, s, a, g);
    else hipLaunchKernelGGL((preprocess_colour_kernel<12, REC_OFF>), grid, block, 0, s, a, g);
}

template <class Body, int MODE>
void launch_preprocess_fwd_mode(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s)
{
    const int shrow = staged_shrow(a);
    const bool sh_in = MODE != PRE_NOCOLOUR && shrow > 0 && 2 * (a.D + 1) * (a.D + 1) >= a.M && aligned16(a.shs);
    const dim3 grid((a.P + 63) / 64), block(64);
    switch (sh_in ? shrow : 0)
    {
    case 48: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 48, true, MODE>), grid, block, 0, s, a, radii, g); break; // rows are 16-byte multiples
    case 27: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 27, false, MODE>), grid, block, 0, s, a, radii, g); break;
    case 12: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 12, true, MODE>), grid, block, 0, s, a, radii, g); break;
    case 3: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 3, false, MODE>), grid, block, 0, s, a, radii, g); break;
    default: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 0, false, MODE>), grid, block, 0, s, a, radii, g); break;
    }
}

template <class Body>
void launch_preprocess_fwd(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s, int mode)
{
    if (a.P <= 0) return;
    if (!aligned16(a.vertex)) // callers ask preprocess_fwd_splittable() first: an unaligned scene only ever comes here with PRE_ALL
    {
        hipLaunchKernelGGL((preprocess_fwd_direct_kernel<Body>), dim3((a.P + 255) / 256), dim3(256), 0, s, a, radii, g);
        return;
    }
    if (mode == PRE_NOCOLOUR) launch_preprocess_fwd_mode<Body, PRE_NOCOLOUR>(a, radii, g, s);
    else launch_preprocess_fwd_mode<Body, PRE_ALL>(a, radii, g, s);
}

#define TS_BWD_STAGED(SHROW, SH_IN, WRITE_SH)                                                                                \
    hipLaunchKernelGGL((preprocess_bwd_staged_kernel<Body, SHROW, SH_IN, WRITE_SH, (SHROW > 0 && SHROW % 4 == 0)>), grid, block, 0, s, a, \
                       radii, g, grad_rec, dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity)
#define TS_BWD_STAGED_ROW(SHROW)                                                                                             \
    do                                                                                                                       \
    {                                                                                                                        \
        if (sh_in && write_sh) TS_BWD_STAGED(SHROW, true, true);                                                             \
        else if (sh_in) TS_BWD_STAGED(SHROW, true, false);                                                                   \
        else TS_BWD_STAGED(SHROW, false, true);                                                                              \
    } while (0)

template <class Body>
void launch_preprocess_bwd(const PreprocessArgs &a, const int32_t *radii, const GeometryStateView &g, const float *grad_rec,
                           float *dL_dvertex, float *dL_dcenter2D, float *dL_dshs, float *dL_dfeature, float *dL_dopacity,
                           hipStream_t s)
{
    if (a.P <= 0) return;
    const int shrow = staged_shrow(a);
    const bool write_sh = a.use_shs && dL_dshs != nullptr;
    const bool ok = aligned16(a.vertex) && aligned16(dL_dvertex) && (!a.use_shs || aligned16(a.shs)) &&
                    (!write_sh || aligned16(dL_dshs));
    const bool sh_in = shrow > 0 && 2 * (a.D + 1) * (a.D + 1) >= a.M;
    const int rows = (sh_in || write_sh) ? shrow : 0; // the LDS rows exist when they carry something
    if (!ok || (rows == 0 && write_sh)) // unaligned, or M outside {1,4,9,16} with dL_dshs to write
    {
        hipLaunchKernelGGL((preprocess_bwd_direct_kernel<Body>), dim3((a.P + 255) / 256), dim3(256), 0, s, a, radii, g, grad_rec,
                           dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity);
        return;
    }
    const dim3 grid((a.P + 63) / 64), block(64);
    switch (rows)
    {
    case 48: TS_BWD_STAGED_ROW(48); break;
    case 27: TS_BWD_STAGED_ROW(27); break;
    case 12: TS_BWD_STAGED_ROW(12); break;
    case 3: TS_BWD_STAGED_ROW(3); break;
    default: TS_BWD_STAGED(0, false, false); break;
    }
}
#undef TS_BWD_STAGED_ROW
#undef TS_BWD_STAGED
} // namespace ts
