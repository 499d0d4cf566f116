// model_update.hip -- model-update operators of the reference's VanillaTSModel as gfx950 kernels (include/ts_model.h):
//   * the per-iteration densification statistics (VanillaTS_model.py:347-363), one fused pass;
//   * the building blocks of the periodic structural updates (:214-345, 365-532): stable row compaction by mask and row
//     gathers by index (pruning, growth, Adam-state surgery), the grow classification + split geometry of `_grow_points`
//     (:260-315), the pruning masks (:384-427), the clipping / reset updates with their Adam-state zeroing (:316-345,
//     :397-409, :446-463, :524-537).
// All of them are HBM-bound one-pass kernels over per-triangle rows; the host mirror (diff_recon_hip/model_update.py) strings
// them together under the reference's method names.
#include "../../include/ts_model.h"
#include "ts2d_common.h"
#include "ts2d_wave.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "mask_count_kernel's block hand-off relies on gfx950's write-through store / L1-bypassing load behaviour; re-validate before building for another target"
#endif
namespace
{
__global__ void __launch_bounds__(256) training_statistic_kernel(int P, int V, const int32_t *__restrict__ radii,
                                                                  const float2 *__restrict__ c2d_grad,
                                                                  const float *__restrict__ csum, const float *__restrict__ cmax,
                                                                  float *__restrict__ g_accum, float *__restrict__ g_denom,
                                                                  float *__restrict__ max_radii, float *__restrict__ s_csum,
                                                                  float *__restrict__ s_cmax, float *__restrict__ c_denom)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float ga = 0.0f, cnt = 0.0f, mr = -1.0f, ms = 0.0f, mm = 0.0f;
    for (int v = 0; v < V; v++)
    {
        const size_t o = (size_t)v * P + i;
        const int r = radii[o];
        if (r <= 0) continue; // visible_mask = radii > 0 (VanillaTS_model.py:679)
        const float2 g = c2d_grad[o];
        ga += sqrtf(g.x * g.x + g.y * g.y); // :358
        cnt += 1.0f;                        // :359, :362
        mr = fmaxf(mr, (float)r);           // :363
        if (csum) { ms = fmaxf(ms, csum[o]); mm = fmaxf(mm, cmax[o]); } // :360-361 (contributions are >= 0)
    }
    if (cnt == 0.0f) return; // invisible in every view: state untouched, like the masked assignments
    g_accum[i] += ga;
    g_denom[i] += cnt;
    c_denom[i] += cnt;
    max_radii[i] = fmaxf(max_radii[i], mr);
    if (csum) { s_csum[i] = fmaxf(s_csum[i], ms); s_cmax[i] = fmaxf(s_cmax[i], mm); }
}

// ---- stable compaction: position of every selected row = number of selected rows before it -------------------------------
// Two launches: per-block counts (+ their exclusive prefix by the block that arrives last), then the positions.
constexpr int MB = 1024; // rows per block

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wtot, uint32_t &block_total)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1)
    {
        const uint32_t y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    __syncthreads();
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += wtot[w];
    block_total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    return base + inc - v;
}

__global__ void __launch_bounds__(256) mask_count_kernel(int P, const uint8_t *__restrict__ mask, uint32_t match, uint32_t *blocksum,
                                                          uint32_t *ticket)
{
    __shared__ uint32_t wtot[4];
    __shared__ bool last;
    const int t = threadIdx.x;
    const int i0 = blockIdx.x * MB + 4 * t;
    uint32_t c = 0;
    for (int k = 0; k < 4; k++)
        if (i0 + k < P && mask[i0 + k] == match) c++;
    uint32_t total;
    (void)block_exclusive_scan(c, wtot, total);
    const int nblocks = gridDim.x;
    if (t == 0)
    {
        // same hand-off as binning.hip's last_arrival(): write-through store, drained, then the ticket; sc1 loads in the elected block --
        // a gfx950 hardware contract (MI355X_MICROARCH.md "valid forms"), checked for the target at the top of this file
        __hip_atomic_store(blocksum + blockIdx.x, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (tk == (uint32_t)nblocks - 1u);
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) return;
    __shared__ uint32_t carry;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 256)
    {
        const int b = b0 + t;
        const uint32_t x = (b < nblocks) ? __hip_atomic_load(blocksum + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan(x, wtot, tot);
        const uint32_t cbase = carry;
        if (b < nblocks) blocksum[b] = cbase + ex;
        __syncthreads();
        if (t == 0) carry = cbase + tot;
        __syncthreads();
    }
    if (t == 0) blocksum[nblocks] = carry; // number of selected rows
}

__global__ void __launch_bounds__(256) mask_positions_kernel(int P, const uint8_t *__restrict__ mask, uint32_t match,
                                                              const uint32_t *__restrict__ blocksum, uint32_t *__restrict__ pos)
{
    __shared__ uint32_t wtot[4];
    const int t = threadIdx.x;
    const int i0 = blockIdx.x * MB + 4 * t;
    bool sel[4];
    uint32_t c = 0;
    for (int k = 0; k < 4; k++)
    {
        sel[k] = i0 + k < P && mask[i0 + k] == match;
        c += sel[k];
    }
    uint32_t total;
    uint32_t p = blocksum[blockIdx.x] + block_exclusive_scan(c, wtot, total);
    for (int k = 0; k < 4; k++)
        if (i0 + k < P)
        {
            pos[i0 + k] = sel[k] ? p : 0xFFFFFFFFu;
            p += sel[k];
        }
}

// dst[pos[i]] = src[i] for the selected rows (pos[i] != ~0); rows are `row_words` 32-bit words
__global__ void __launch_bounds__(256) scatter_rows_kernel(int64_t total_words, int row_words, const uint32_t *__restrict__ pos,
                                                            const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t dst_row0)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= total_words) return;
    const int64_t row = w / row_words;
    const int col = (int)(w - row * row_words);
    const uint32_t p = pos[row];
    if (p != 0xFFFFFFFFu) dst[(dst_row0 + p) * row_words + col] = src[w];
}
// dst[dst_row0 + j] = src[idx[j]]
__global__ void __launch_bounds__(256) gather_rows_kernel(int64_t total_words, int row_words, const uint32_t *__restrict__ idx,
                                                           const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int64_t dst_row0)
{
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= total_words) return;
    const int64_t j = w / row_words;
    const int col = (int)(w - j * row_words);
    dst[(dst_row0 + j) * row_words + col] = src[(int64_t)idx[j] * row_words + col];
}

__device__ __forceinline__ float side_len(const float *a, const float *b)
{
    const float x = a[0] - b[0], y = a[1] - b[1], z = a[2] - b[2];
    return sqrtf(x * x + y * y + z * z);
}
// get_scaling (VanillaTS_model.py:72-76): mean side length, sides in the order (v3 - v2, v1 - v3, v2 - v1)
__device__ __forceinline__ float mean_side(const float *v, float &l1, float &l2, float &l3)
{
    l1 = side_len(v + 6, v + 3);
    l2 = side_len(v + 0, v + 6);
    l3 = side_len(v + 3, v + 0);
    return (l1 + l2 + l3) / 3.0f;
}

// _densification (:365-383) + the classification of _grow_points (:260-263): code 0 = untouched, 1 = clone, 2 = split;
// gradient_accum / gradient_denom of the selected triangles are reset
__global__ void __launch_bounds__(256) grow_classify_kernel(int P, const float *__restrict__ vertex, float *__restrict__ g_accum,
                                                             float *__restrict__ g_denom, float min_view_count, float grad_threshold,
                                                             float split_scale_threshold, uint8_t *__restrict__ code)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float den = g_denom[i], acc = g_accum[i];
    const bool select = den >= min_view_count;              // :376
    const bool grow = select && acc > grad_threshold * den; // :377-379
    float l1, l2, l3;
    const bool large = mean_side(vertex + 9 * (size_t)i, l1, l2, l3) > split_scale_threshold; // :261
    code[i] = grow ? (large ? 2 : 1) : 0;
    if (select) { g_accum[i] = 0.0f; g_denom[i] = 0.0f; } // :381-382
}

// the two children of a split triangle (:270-283): the longest side is cut at its centre
__global__ void __launch_bounds__(256) split_vertex_kernel(int n_split, const uint32_t *__restrict__ parents, const float *__restrict__ vertex,
                                                            float *__restrict__ child1, float *__restrict__ child2)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_split) return;
    const float *v = vertex + 9 * (size_t)parents[j];
    float l1, l2, l3;
    (void)mean_side(v, l1, l2, l3);
    int l = 0; // torch.argmax: first maximum
    if (l2 > l1) l = 1;
    if (l3 > fmaxf(l1, l2)) l = 2;
    const int p1 = (l + 1) % 3, p2 = (l + 2) % 3;
    float c[3];
    for (int k = 0; k < 3; k++) c[k] = (v[3 * p1 + k] + v[3 * p2 + k]) / 2.0f; // :278
    float *a = child1 + 9 * (size_t)j, *b = child2 + 9 * (size_t)j;
    for (int k = 0; k < 3; k++)
    {
        a[k] = v[3 * l + k]; a[3 + k] = v[3 * p1 + k]; a[6 + k] = c[k]; // :279
        b[k] = v[3 * l + k]; b[3 + k] = c[k]; b[6 + k] = v[3 * p2 + k]; // :280
    }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// pruning masks: mode 0 = opacity (sigmoid(opacity) < threshold, :391), 1 = opacity clipping mask (sigmoid(opacity) > threshold, :403),
// 2 = scale pruning (max_radii2D > a || mean side > b, :417-419), 3 = scale clipping mask (mean side > a, :453)
__global__ void __launch_bounds__(256) update_mask_kernel(int P, int mode, const float *__restrict__ opacity, const float *__restrict__ vertex,
                                                           const float *__restrict__ max_radii, float a, float b, uint8_t *__restrict__ mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    bool m = false;
    float l1, l2, l3;
    if (mode == 0) m = sigmoidf(opacity[i]) < a;
    else if (mode == 1) m = sigmoidf(opacity[i]) > a;
    else if (mode == 2) m = max_radii[i] > a || mean_side(vertex + 9 * (size_t)i, l1, l2, l3) > b;
    else m = mean_side(vertex + 9 * (size_t)i, l1, l2, l3) > a;
    mask[i] = m ? 1 : 0;
}

// _clipping_update_states (:330-345) for the masked rows: parameter row <- value (opacity: the constant; vertex: rescaled about the
// centre so that the mean side becomes scale_max, :429-463), Adam moments <- 0
__global__ void __launch_bounds__(256) clip_kernel(int P, int mode, const uint8_t *__restrict__ mask, float value, float *__restrict__ param,
                                                    float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || !mask[i]) return;
    if (mode == 0)
    {
        param[i] = value;
        if (exp_avg) { exp_avg[i] = 0.0f; exp_avg_sq[i] = 0.0f; }
        return;
    }
    float *v = param + 9 * (size_t)i;
    float l1, l2, l3;
    const float ratio = value / mean_side(v, l1, l2, l3); // :456
    for (int k = 0; k < 3; k++)
    {
        const float c = (v[k] + v[3 + k] + v[6 + k]) / 3.0f; // mean over the three vertices (:443)
        for (int j = 0; j < 3; j++) v[3 * j + k] = (v[3 * j + k] - c) * ratio + c; // :444
    }
    if (exp_avg)
        for (int k = 0; k < 9; k++) { exp_avg[9 * (size_t)i + k] = 0.0f; exp_avg_sq[9 * (size_t)i + k] = 0.0f; }
}

// _opacity_reset (:524-537): opacity <- inverse_sigmoid(min(sigmoid(opacity), reset_value)), Adam moments of every row <- 0
__global__ void __launch_bounds__(256) opacity_reset_kernel(int P, float reset_value, float *__restrict__ opacity, float *__restrict__ exp_avg,
                                                             float *__restrict__ exp_avg_sq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float x = fminf(sigmoidf(opacity[i]), reset_value);
    opacity[i] = logf(x / (1.0f - x)); // inverse_sigmoid, model_utils.py
    if (exp_avg) { exp_avg[i] = 0.0f; exp_avg_sq[i] = 0.0f; }
}
// bg_depth of VanillaTSModel.forward (:623): max over all vertices of |camera_center - vertex|.  torch spends three kernels on it (subtract, norm, max:
// 61 us per view at 1 M triangles, profiles/r06_train_step_kernels.txt); this is one read of the vertices.  Distances are >= 0, so the unsigned bit
// pattern orders like the value and one atomicMax per block lands the result (`out` zeroed by max_distance_zero_kernel in front).
__global__ void max_distance_zero_kernel(uint32_t *__restrict__ out) { out[0] = 0u; }

__global__ void __launch_bounds__(256) max_distance_kernel(int n, const float *__restrict__ vertex, const float *__restrict__ campos,
                                                           uint32_t *__restrict__ out)
{
    __shared__ float wmax[4];
    const float cx = campos[0], cy = campos[1], cz = campos[2];
    float m = 0.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    {
        const float dx = cx - vertex[3 * (size_t)i], dy = cy - vertex[3 * (size_t)i + 1], dz = cz - vertex[3 * (size_t)i + 2];
        m = fmaxf(m, __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out, __float_as_uint(sqrtf(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])))));
}
} // namespace

hipError_t ts_model_training_statistic(int P, int V, const int32_t *radii, const float *c2d_grad, const float *csum, const float *cmax,
                                       float *g_accum, float *g_denom, float *max_radii, float *s_csum, float *s_cmax, float *c_denom,
                                       hipStream_t s)
{
    if (P <= 0 || V <= 0) return hipSuccess;
    hipLaunchKernelGGL(training_statistic_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, V, radii, (const float2 *)c2d_grad, csum,
                       cmax, g_accum, g_denom, max_radii, s_csum, s_cmax, c_denom);
    return hipGetLastError();
}

size_t ts_model_select_scratch_bytes(int P) { return ((size_t)(P > 0 ? (P + MB - 1) / MB : 0) + 8) * sizeof(uint32_t); }

hipError_t ts_model_select_rows(int P, const uint8_t *mask, int match, uint32_t *pos, uint32_t *scratch, uint32_t *count_host, hipStream_t s)
{
    *count_host = 0;
    if (P <= 0) return hipSuccess;
    const int nblocks = (P + MB - 1) / MB;
    uint32_t *ticket = scratch + nblocks + 1;
    hipError_t e = hipMemsetAsync(ticket, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mask_count_kernel, dim3(nblocks), dim3(256), 0, s, P, mask, (uint32_t)match, scratch, ticket);
    hipLaunchKernelGGL(mask_positions_kernel, dim3(nblocks), dim3(256), 0, s, P, mask, (uint32_t)match, scratch, pos);
    e = hipMemcpyAsync(count_host, scratch + nblocks, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return e == hipSuccess ? hipGetLastError() : e;
}

hipError_t ts_model_scatter_rows(int64_t rows, int row_words, const uint32_t *pos, const void *src, void *dst, int64_t dst_row0, hipStream_t s)
{
    const int64_t words = rows * row_words;
    if (words <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, words, row_words, pos, (const uint32_t *)src,
                       (uint32_t *)dst, dst_row0);
    return hipGetLastError();
}

hipError_t ts_model_gather_rows(int64_t rows, int row_words, const uint32_t *idx, const void *src, void *dst, int64_t dst_row0, hipStream_t s)
{
    const int64_t words = rows * row_words;
    if (words <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, words, row_words, idx, (const uint32_t *)src,
                       (uint32_t *)dst, dst_row0);
    return hipGetLastError();
}

hipError_t ts_model_grow_classify(int P, const float *vertex, float *g_accum, float *g_denom, float min_view_count, float grad_threshold,
                                  float split_scale_threshold, uint8_t *code, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(grow_classify_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, vertex, g_accum, g_denom, min_view_count, grad_threshold,
                       split_scale_threshold, code);
    return hipGetLastError();
}

hipError_t ts_model_split_vertex(int n_split, const uint32_t *parents, const float *vertex, float *child1, float *child2, hipStream_t s)
{
    if (n_split <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_vertex_kernel, dim3((n_split + 255) / 256), dim3(256), 0, s, n_split, parents, vertex, child1, child2);
    return hipGetLastError();
}

hipError_t ts_model_update_mask(int P, int mode, const float *opacity, const float *vertex, const float *max_radii, float a, float b, uint8_t *mask,
                                hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(update_mask_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, mode, opacity, vertex, max_radii, a, b, mask);
    return hipGetLastError();
}

hipError_t ts_model_clip(int P, int mode, const uint8_t *mask, float value, float *param, float *exp_avg, float *exp_avg_sq, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(clip_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, mode, mask, value, param, exp_avg, exp_avg_sq);
    return hipGetLastError();
}

hipError_t ts_model_opacity_reset(int P, float reset_value, float *opacity, float *exp_avg, float *exp_avg_sq, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    hipLaunchKernelGGL(opacity_reset_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, reset_value, opacity, exp_avg, exp_avg_sq);
    return hipGetLastError();
}

hipError_t ts_model_max_distance(int n_vertices, const float *vertex, const float *campos, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(max_distance_zero_kernel, dim3(1), dim3(1), 0, s, (uint32_t *)out);
    if (n_vertices > 0)
    {
        const int blocks = (n_vertices + 255) / 256;
        hipLaunchKernelGGL(max_distance_kernel, dim3(blocks < 2048 ? blocks : 2048), dim3(256), 0, s, n_vertices, vertex, campos, (uint32_t *)out);
    }
    return hipGetLastError();
}
