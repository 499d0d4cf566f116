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

namespace ts
{
// The binning kernels' "last block finishes" tickets (binning.hip) are zeroed by the first launch of the step.  Every grid has at
// least 64 threads and slabs + 8 <= max(P, 64), so the threads beyond P of a tiny scene take part.
__device__ __forceinline__ void clear_tickets(const GeometryStateView &g, int idx)
{
    if (idx < g.rs.slabs + 8) g.rs.tickets[idx] = 0u;
}

template <class Body>
__global__ void __launch_bounds__(256) preprocess_fwd_direct_kernel(PreprocessArgs a, int32_t *__restrict__ radii, GeometryStateView g)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    clear_tickets(g, idx);
    if (idx >= a.P) return;
    Body::fwd(a, radii, g, idx, a.vertex + 9 * (size_t)idx, a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr);
}

// vertex rows always staged; SH rows staged when SHROW = 3 M > 0
template <class Body, int SHROW>
__global__ void __launch_bounds__(64) preprocess_fwd_staged_kernel(PreprocessArgs a, int32_t *__restrict__ radii, GeometryStateView g)
{
    __shared__ float s_v[64 * 9];
    __shared__ float s_sh[SHROW > 0 ? 64 * (SHROW + 1) : 1];
    const int lane = threadIdx.x, row0 = blockIdx.x * 64, idx = row0 + lane;
    stage_rows_in<9, 9>(s_v, a.vertex, row0, a.P, lane);
    if (SHROW > 0) stage_rows_in<SHROW, SHROW + 1>(s_sh, a.shs, row0, a.P, lane);
    __syncthreads();
    clear_tickets(g, idx);
    if (idx >= a.P) return;
    const float *shp = SHROW > 0 ? s_sh + lane * (SHROW + 1) : (a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr);
    Body::fwd(a, radii, g, idx, s_v + lane * 9, shp);
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
template <class Body, int SHROW, bool SH_IN, bool WRITE_SH>
__global__ void __launch_bounds__(64) preprocess_bwd_staged_kernel(PreprocessArgs a, const int32_t *__restrict__ radii,
                                                                    GeometryStateView g, const float *__restrict__ grad_rec,
                                                                    float *__restrict__ dL_dvertex, float *__restrict__ dL_dcenter2D,
                                                                    float *__restrict__ dL_dshs, float *__restrict__ dL_dfeature,
                                                                    float *__restrict__ dL_dopacity)
{
    __shared__ float s_v[64 * 9];
    __shared__ float s_sh[SHROW > 0 ? 64 * (SHROW + 1) : 1];
    const int lane = threadIdx.x, row0 = blockIdx.x * 64, idx = row0 + lane;
    stage_rows_in<9, 9>(s_v, a.vertex, row0, a.P, lane);
    if (SHROW > 0 && SH_IN) stage_rows_in<SHROW, SHROW + 1>(s_sh, a.shs, row0, a.P, lane);
    __syncthreads();
    if (idx < a.P)
    {
        float *row = SHROW > 0 ? s_sh + lane * (SHROW + 1) : nullptr;
        const float *shp = (SHROW > 0 && SH_IN) ? row : (a.use_shs ? a.shs + (size_t)idx * a.M * 3 : nullptr);
        Body::bwd(a, radii, g, grad_rec, idx, s_v + lane * 9, shp, s_v + lane * 9, WRITE_SH ? row : nullptr, dL_dcenter2D,
                  dL_dfeature, dL_dopacity);
    }
    __syncthreads();
    stage_rows_out<9, 9>(s_v, dL_dvertex, row0, a.P, lane);
    if (SHROW > 0 && WRITE_SH) stage_rows_out<SHROW, SHROW + 1>(s_sh, dL_dshs, row0, a.P, lane);
}

// Staging policy: vertex rows whenever the pointers are 16-byte aligned; SH rows in when at least half of each row is
// active (otherwise the direct strided read of the active prefix moves fewer bytes); dL_dshs rows out always (every
// element is written).  M outside {1, 4, 9, 16} never stages SH rows.
static inline int staged_shrow(const PreprocessArgs &a) { return (a.use_shs && (a.M == 1 || a.M == 4 || a.M == 9 || a.M == 16)) ? 3 * a.M : 0; }

template <class Body>
void launch_preprocess_fwd(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s)
{
    if (a.P <= 0) return;
    const int shrow = staged_shrow(a);
    const bool sh_in = shrow > 0 && 2 * (a.D + 1) * (a.D + 1) >= a.M && aligned16(a.shs);
    if (!aligned16(a.vertex))
    {
        hipLaunchKernelGGL((preprocess_fwd_direct_kernel<Body>), dim3((a.P + 255) / 256), dim3(256), 0, s, a, radii, g);
        return;
    }
    const dim3 grid((a.P + 63) / 64), block(64);
    switch (sh_in ? shrow : 0)
    {
    case 48: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 48>), grid, block, 0, s, a, radii, g); break;
    case 27: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 27>), grid, block, 0, s, a, radii, g); break;
    case 12: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 12>), grid, block, 0, s, a, radii, g); break;
    case 3: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 3>), grid, block, 0, s, a, radii, g); break;
    default: hipLaunchKernelGGL((preprocess_fwd_staged_kernel<Body, 0>), grid, block, 0, s, a, radii, g); break;
    }
}

#define TS_BWD_STAGED(SHROW, SH_IN, WRITE_SH)                                                                                \
    hipLaunchKernelGGL((preprocess_bwd_staged_kernel<Body, SHROW, SH_IN, WRITE_SH>), grid, block, 0, s, a, radii, g, grad_rec, \
                       dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity)
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
