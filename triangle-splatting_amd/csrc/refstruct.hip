// refstruct.hip -- MEASUREMENT AID, not the product path.
//
// BASELINE.md section 3 asks for a "structure-faithful HIP baseline mode" as the only available proxy for "the
// reference algorithm on this hardware" (the reference's CUDA cannot run here and publishes no numbers): from-scratch
// HIP kernels that keep the STRUCTURE of FORWARD::renderCUDA / BACKWARD::renderCUDA (R2D/src/forward.cu:198-355,
// R2D/src/backward.cu:265-493): one 16x16 thread block per tile, one thread per pixel, the tile's list streamed
// through shared memory in 256-entry batches with two barriers per batch and a block-wide "all done" vote, and one
// global float atomic per (pixel, triangle, value) -- 2 in the forward (contrib_sum / contrib_max), 16 in the
// backward.  Selected with the environment variable TS2D_MODE=refstruct (bench.py --mode refstruct); results are the
// same as the product kernels' up to float summation order (tests/test_parity_gpu.py::test_refstruct_mode_matches).
// It reuses this library's 64-byte render / gradient records, binning and preprocess kernels, so the comparison
// isolates the blend-kernel structure.
#include "ts2d_common.h"

namespace
{
constexpr int BATCH = 256;

struct Staged // one list entry staged in LDS
{
    float v1x, v1y, v2x, v2y, v3x, v3y, inv_area, op;
    float r, g, b, nx, ny, nz, d1, d2, d3;
    uint32_t id;
};

__device__ __forceinline__ float pow_g(float x, float y) { return y == 2.0f ? x * x : powf(x, y); }

template <bool RICH>
__global__ void __launch_bounds__(256) refstruct_fwd_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                             const uint32_t *__restrict__ point_list,
                                                             const float4 *__restrict__ rec, float *__restrict__ final_T,
                                                             uint32_t *__restrict__ n_contrib, float *__restrict__ out_feature,
                                                             float *__restrict__ out_depth, float *__restrict__ out_normal,
                                                             float *__restrict__ contrib_sum, float *__restrict__ contrib_max)
{
    __shared__ Staged st[BATCH];
    __shared__ int votes;
    const int tile = blockIdx.x, tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = tx * TS_TILE + lx, py = ty * TS_TILE + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)px, fy = (float)py;
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    const float g2 = 2.0f * a.gamma;
    float T = 1.0f, ar = 0, ag = 0, ab = 0, anx = 0, any_ = 0, anz = 0, ad = 0;
    bool done = !inside;
    uint32_t contributor = 0, last = 0;
    for (int base = 0; base < len; base += BATCH)
    {
        if (threadIdx.x == 0) votes = 0;
        __syncthreads();
        if (done) atomicAdd(&votes, 1);
        __syncthreads();
        if (votes == BATCH) break; // __syncthreads_count(done) == BLOCK_SIZE, forward.cu:265-267
        const int k = base + threadIdx.x;
        if (k < len)
        {
            const uint32_t id = point_list[range.x + k];
            const float4 *rp = rec + 4 * (size_t)id;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = RICH ? rp[3] : make_float4(0, 0, 0, 0);
            Staged s;
            s.v1x = r0.x; s.v1y = r0.y; s.v2x = r0.z; s.v2y = r0.w; s.v3x = r1.x; s.v3y = r1.y;
            s.inv_area = 1.0f / ((r0.z - r0.x) * (r1.y - r0.y) - (r0.w - r0.y) * (r1.x - r0.x));
            s.op = r1.z; s.r = r1.w; s.g = r2.x; s.b = r2.y; s.nx = r2.z; s.ny = r2.w; s.nz = r3.x;
            s.d1 = r3.y; s.d2 = r3.z; s.d3 = r3.w; s.id = id;
            st[threadIdx.x] = s;
        }
        __syncthreads();
        const int cnt = min(BATCH, len - base);
        for (int j = 0; !done && j < cnt; j++)
        {
            contributor++;
            last = contributor;
            const Staged &s = st[j];
            const float p1x = s.v1x - fx, p1y = s.v1y - fy, p2x = s.v2x - fx, p2y = s.v2y - fy, p3x = s.v3x - fx, p3y = s.v3y - fy;
            const float a1 = (p2x * p3y - p2y * p3x) * s.inv_area, a2 = (p3x * p1y - p3y * p1x) * s.inv_area, a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            if (ecc < 0.0f || ecc > 10.0f) continue;
            const float alpha = fminf(0.99f, s.op * __expf(-0.5f * pow_g(ecc, g2)));
            if (alpha < 1.0f / 255.0f) continue;
            const float contrib = alpha * T;
            ar += s.r * contrib; ag += s.g * contrib; ab += s.b * contrib;
            if (RICH)
            {
                unsafeAtomicAdd(contrib_sum + s.id, contrib);                         // forward.cu:323
                atomicMax((int *)contrib_max + s.id, __float_as_int(contrib));        // forward.cu:324
                anx += s.nx * contrib; any_ += s.ny * contrib; anz += s.nz * contrib;
                ad += (s.d1 * a1 + s.d2 * a2 + s.d3 * a3) * contrib;
            }
            T *= (1.0f - alpha);
            if (T <= 0.0001f) done = true;
        }
    }
    if (inside)
    {
        const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_feature[pix] = ar + T * a.background[0];
        if (a.C > 1) out_feature[HW + pix] = ag + T * a.background[1];
        if (a.C > 2) out_feature[2 * HW + pix] = ab + T * a.background[2];
        if (RICH)
        {
            out_depth[pix] = ad + T * a.background_depth;
            out_normal[pix] = anx; out_normal[HW + pix] = any_; out_normal[2 * HW + pix] = anz;
        }
    }
}

template <bool RICH>
__global__ void __launch_bounds__(256) refstruct_bwd_kernel(RenderArgs a, const uint2 *__restrict__ ranges,
                                                             const uint32_t *__restrict__ point_list,
                                                             const float4 *__restrict__ rec, const float *__restrict__ final_T,
                                                             const uint32_t *__restrict__ n_contrib,
                                                             const float *__restrict__ dL_dout_feature,
                                                             const float *__restrict__ dL_dout_depth,
                                                             const float *__restrict__ dL_dout_normal, float *__restrict__ grad_rec)
{
    __shared__ Staged st[BATCH];
    const int tile = blockIdx.x, tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int px = tx * TS_TILE + lx, py = ty * TS_TILE + ly;
    const bool inside = px < a.W && py < a.H;
    const float fx = (float)px, fy = (float)py;
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    const float g2 = 2.0f * a.gamma;
    const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
    float T = inside ? final_T[pix] : 0.0f;
    const uint32_t last = inside ? n_contrib[pix] : 0;
    uint32_t contributor = (uint32_t)len;
    float acr = 0, acg = 0, acb = 0, acnx = 0, acny = 0, acnz = 0, acd = a.background_depth;
    float dpr = 0, dpg = 0, dpb = 0, dnx = 0, dny = 0, dnz = 0, dd = 0;
    if (inside)
    {
        acr = a.background[0]; dpr = dL_dout_feature[pix];
        if (a.C > 1) { acg = a.background[1]; dpg = dL_dout_feature[HW + pix]; }
        if (a.C > 2) { acb = a.background[2]; dpb = dL_dout_feature[2 * HW + pix]; }
        if (RICH) { dnx = dL_dout_normal[pix]; dny = dL_dout_normal[HW + pix]; dnz = dL_dout_normal[2 * HW + pix]; dd = dL_dout_depth[pix]; }
    }
    for (int base = 0; base < len; base += BATCH)
    {
        __syncthreads();
        const int k = base + threadIdx.x;
        if (k < len)
        {
            const uint32_t id = point_list[range.y - 1 - k]; // back to front, backward.cu:353
            const float4 *rp = rec + 4 * (size_t)id;
            const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = RICH ? rp[3] : make_float4(0, 0, 0, 0);
            Staged s;
            s.v1x = r0.x; s.v1y = r0.y; s.v2x = r0.z; s.v2y = r0.w; s.v3x = r1.x; s.v3y = r1.y;
            s.inv_area = 1.0f / ((r0.z - r0.x) * (r1.y - r0.y) - (r0.w - r0.y) * (r1.x - r0.x));
            s.op = r1.z; s.r = r1.w; s.g = r2.x; s.b = r2.y; s.nx = r2.z; s.ny = r2.w; s.nz = r3.x;
            s.d1 = r3.y; s.d2 = r3.z; s.d3 = r3.w; s.id = id;
            st[threadIdx.x] = s;
        }
        __syncthreads();
        const int cnt = min(BATCH, len - base);
        for (int j = 0; inside && j < cnt; j++)
        {
            contributor--;
            if (contributor >= last) continue;
            const Staged &s = st[j];
            const float p1x = s.v1x - fx, p1y = s.v1y - fy, p2x = s.v2x - fx, p2y = s.v2y - fy, p3x = s.v3x - fx, p3y = s.v3y - fy;
            const float a1 = (p2x * p3y - p2y * p3x) * s.inv_area, a2 = (p3x * p1y - p3y * p1x) * s.inv_area, a3 = 1.0f - a1 - a2;
            const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
            if (ecc < 0.0f || ecc > 10.0f) continue;
            const float power = -0.5f * pow_g(ecc, g2);
            const float G = __expf(power);
            const float alpha = fminf(0.99f, s.op * G);
            if (alpha < 1.0f / 255.0f) continue;
            T = T / (1.0f - alpha);
            const float contrib = alpha * T, oma = 1.0f - alpha;
            float *g = grad_rec + TS_GRAD_FLOATS * (size_t)s.id;
            float dL_dcontrib = 0.0f, da1 = 0.0f, da2 = 0.0f, da3 = 0.0f;
            unsafeAtomicAdd(g + 7, dpr * contrib); unsafeAtomicAdd(g + 8, dpg * contrib); unsafeAtomicAdd(g + 9, dpb * contrib);
            dL_dcontrib += dpr * (s.r - acr) + dpg * (s.g - acg) + dpb * (s.b - acb);
            acr = alpha * s.r + oma * acr; acg = alpha * s.g + oma * acg; acb = alpha * s.b + oma * acb;
            if (RICH)
            {
                unsafeAtomicAdd(g + 10, dnx * contrib); unsafeAtomicAdd(g + 11, dny * contrib); unsafeAtomicAdd(g + 12, dnz * contrib);
                dL_dcontrib += dnx * (s.nx - acnx) + dny * (s.ny - acny) + dnz * (s.nz - acnz);
                acnx = alpha * s.nx + oma * acnx; acny = alpha * s.ny + oma * acny; acnz = alpha * s.nz + oma * acnz;
                const float dL_ddepth = dd * contrib;
                unsafeAtomicAdd(g + 13, dL_ddepth * a1); unsafeAtomicAdd(g + 14, dL_ddepth * a2); unsafeAtomicAdd(g + 15, dL_ddepth * a3);
                da1 = dL_ddepth * s.d1; da2 = dL_ddepth * s.d2; da3 = dL_ddepth * s.d3;
                const float depth = s.d1 * a1 + s.d2 * a2 + s.d3 * a3;
                dL_dcontrib += dd * (depth - acd);
                acd = alpha * depth + oma * acd;
            }
            const float dL_dalpha = dL_dcontrib * T;
            const float dL_dpower = (s.op * G < 0.99f) ? dL_dalpha * alpha : 0.0f;
            const float dL_decc = dL_dpower * g2 * power / (ecc + 1e-8f);
            if (a1 <= a2 && a1 <= a3) da1 += -3.0f * dL_decc;
            else if (a2 <= a1 && a2 <= a3) da2 += -3.0f * dL_decc;
            else da3 += -3.0f * dL_decc;
            // backward.cu:464-479
            const float e12x = s.v2x - s.v1x, e12y = s.v2y - s.v1y, e23x = s.v3x - s.v2x, e23y = s.v3y - s.v2y;
            const float e31x = s.v1x - s.v3x, e31y = s.v1y - s.v3y, ia = s.inv_area;
            auto perpx = [](float x, float y) { return y; };
            auto perpy = [](float x, float y) { return -x; };
            const float g1x = (da1 * perpx(e23x * a1, e23y * a1) + da2 * perpx(e23x * a2 - p3x, e23y * a2 - p3y) + da3 * perpx(e23x * a3 + p2x, e23y * a3 + p2y)) * ia;
            const float g1y = (da1 * perpy(e23x * a1, e23y * a1) + da2 * perpy(e23x * a2 - p3x, e23y * a2 - p3y) + da3 * perpy(e23x * a3 + p2x, e23y * a3 + p2y)) * ia;
            const float g2x = (da1 * perpx(e31x * a1 + p3x, e31y * a1 + p3y) + da2 * perpx(e31x * a2, e31y * a2) + da3 * perpx(e31x * a3 - p1x, e31y * a3 - p1y)) * ia;
            const float g2y = (da1 * perpy(e31x * a1 + p3x, e31y * a1 + p3y) + da2 * perpy(e31x * a2, e31y * a2) + da3 * perpy(e31x * a3 - p1x, e31y * a3 - p1y)) * ia;
            const float g3x = (da1 * perpx(e12x * a1 - p2x, e12y * a1 - p2y) + da2 * perpx(e12x * a2 + p1x, e12y * a2 + p1y) + da3 * perpx(e12x * a3, e12y * a3)) * ia;
            const float g3y = (da1 * perpy(e12x * a1 - p2x, e12y * a1 - p2y) + da2 * perpy(e12x * a2 + p1x, e12y * a2 + p1y) + da3 * perpy(e12x * a3, e12y * a3)) * ia;
            unsafeAtomicAdd(g + 0, g1x); unsafeAtomicAdd(g + 1, g1y); unsafeAtomicAdd(g + 2, g2x);
            unsafeAtomicAdd(g + 3, g2y); unsafeAtomicAdd(g + 4, g3x); unsafeAtomicAdd(g + 5, g3y);
            unsafeAtomicAdd(g + 6, dL_dalpha * G); // backward.cu:490
        }
    }
}
} // namespace

void ts_launch_refstruct_fwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                             const ImageStateView &im, float *out_feature, float *out_depth, float *out_normal,
                             float *contrib_sum, float *contrib_max, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    if (a.rich_info)
        hipLaunchKernelGGL(refstruct_fwd_kernel<true>, grid, dim3(256), 0, s, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib,
                           out_feature, out_depth, out_normal, contrib_sum, contrib_max);
    else
        hipLaunchKernelGGL(refstruct_fwd_kernel<false>, grid, dim3(256), 0, s, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib,
                           out_feature, out_depth, out_normal, contrib_sum, contrib_max);
}

void ts_launch_refstruct_bwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                             const ImageStateView &im, const float *dL_dout_feature, const float *dL_dout_depth,
                             const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    if (a.rich_info)
        hipLaunchKernelGGL(refstruct_bwd_kernel<true>, grid, dim3(256), 0, s, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib,
                           dL_dout_feature, dL_dout_depth, dL_dout_normal, grad_rec);
    else
        hipLaunchKernelGGL(refstruct_bwd_kernel<false>, grid, dim3(256), 0, s, a, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib,
                           dL_dout_feature, dL_dout_depth, dL_dout_normal, grad_rec);
}
