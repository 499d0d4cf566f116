// knn.hip -- exact 3-NN mean squared distance and nearest-other-group search (include/ts_knn.h).
//
// Reference: submodules/simple-knn/simple_knn.cu ("SK"): bounding box (:241-254, min/max reductions seeded with the
// ORIGIN, a quirk kept here because it shapes the Morton grid and with it the tie order), 30-bit Morton codes (:49-73),
// radix sort of (code, index) (:256-266), min/max box per 1024 sorted points (:82-121, :268-271), then one thread per
// point scanning every box that can still hold a closer point (:153-235).
// Here the sorted points are gathered once into a float4 array (xyz + original index) so every later access is
// contiguous; one 256-lane workgroup owns a box (4 points per lane), scans its own box first to get a search radius,
// and then visits only boxes whose box-to-box distance is within the workgroup's current worst radius, staging each
// visited box through LDS (all lanes read the same candidate: LDS broadcast, no bank conflicts) and skipping per lane
// when the box cannot improve that lane's points.  No host synchronisation anywhere.
#include "../../include/ts_knn.h"
#include "ts2d_common.h"

#include <cfloat>
#include <cstring>

namespace
{
constexpr int BOX = 1024; // SK/auxiliary.h:3
constexpr int TPB = 256, PPT = BOX / TPB;

struct Box { float mnx, mny, mnz, mxx, mxy, mxz; };

__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz)
{
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz; // SK:43-47
}

// SK:123-133
__device__ __forceinline__ float dist_box_point(const Box &b, float px, float py, float pz)
{
    float dx = 0, dy = 0, dz = 0;
    if (px < b.mnx || px > b.mxx) dx = fminf(fabsf(px - b.mnx), fabsf(px - b.mxx));
    if (py < b.mny || py > b.mxy) dy = fminf(fabsf(py - b.mny), fabsf(py - b.mxy));
    if (pz < b.mnz || pz > b.mxz) dz = fminf(fabsf(pz - b.mnz), fabsf(pz - b.mxz));
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ float dist_box_box(const Box &a, const Box &b)
{
    const float dx = fmaxf(0.0f, fmaxf(a.mnx - b.mxx, b.mnx - a.mxx));
    const float dy = fmaxf(0.0f, fmaxf(a.mny - b.mxy, b.mny - a.mxy));
    const float dz = fmaxf(0.0f, fmaxf(a.mnz - b.mxz, b.mnz - a.mxz));
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ float block_reduce(float v, float *red, bool is_max)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : fminf(v, w);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < TPB / 64; i++) r = is_max ? fmaxf(r, red[i]) : fminf(r, red[i]);
    return r;
}

// per-block partial bounding boxes of the raw points
__global__ void __launch_bounds__(TPB) bbox_partial_kernel(int P, const float *__restrict__ pts, Box *__restrict__ partial)
{
    __shared__ float red[TPB / 64];
    Box me = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * TPB + threadIdx.x; i < P; i += gridDim.x * TPB)
    {
        const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        me.mnx = fminf(me.mnx, x); me.mny = fminf(me.mny, y); me.mnz = fminf(me.mnz, z);
        me.mxx = fmaxf(me.mxx, x); me.mxy = fmaxf(me.mxy, y); me.mxz = fmaxf(me.mxz, z);
    }
    Box r;
    r.mnx = block_reduce(me.mnx, red, false); r.mny = block_reduce(me.mny, red, false); r.mnz = block_reduce(me.mnz, red, false);
    r.mxx = block_reduce(me.mxx, red, true); r.mxy = block_reduce(me.mxy, red, true); r.mxz = block_reduce(me.mxz, red, true);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ void __launch_bounds__(64) bbox_finish_kernel(int n, const Box *__restrict__ partial, Box *__restrict__ out)
{
    // the reductions of the reference start from init = {0, 0, 0} (SK:240,245,249): the box always contains the origin
    Box me = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 64)
    {
        const Box b = partial[i];
        me.mnx = fminf(me.mnx, b.mnx); me.mny = fminf(me.mny, b.mny); me.mnz = fminf(me.mnz, b.mnz);
        me.mxx = fmaxf(me.mxx, b.mxx); me.mxy = fmaxf(me.mxy, b.mxy); me.mxz = fmaxf(me.mxz, b.mxz);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
    {
        me.mnx = fminf(me.mnx, __shfl_xor(me.mnx, o)); me.mny = fminf(me.mny, __shfl_xor(me.mny, o));
        me.mnz = fminf(me.mnz, __shfl_xor(me.mnz, o)); me.mxx = fmaxf(me.mxx, __shfl_xor(me.mxx, o));
        me.mxy = fmaxf(me.mxy, __shfl_xor(me.mxy, o)); me.mxz = fmaxf(me.mxz, __shfl_xor(me.mxz, o));
    }
    if (threadIdx.x == 0) *out = me;
}

__device__ __forceinline__ uint32_t prep_morton(uint32_t x) // SK:49-56
{
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}

__device__ __forceinline__ uint32_t quantise(float v, float lo, float hi)
{
    const float t = ((v - lo) / (hi - lo)) * 1023.0f; // SK:60
    return (t >= 0.0f) ? (uint32_t)fminf(t, 4294967040.0f) : 0u; // NaN / negative -> 0 (CUDA float->uint saturation)
}

__global__ void __launch_bounds__(TPB) morton_kernel(int P, const float *__restrict__ pts, const Box *__restrict__ bb,
                                                      uint32_t *__restrict__ codes, uint32_t *__restrict__ ids)
{
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= P) return;
    const Box b = *bb;
    const uint32_t x = prep_morton(quantise(pts[3 * (size_t)i], b.mnx, b.mxx));
    const uint32_t y = prep_morton(quantise(pts[3 * (size_t)i + 1], b.mny, b.mxy));
    const uint32_t z = prep_morton(quantise(pts[3 * (size_t)i + 2], b.mnz, b.mxz));
    codes[i] = x | (y << 1) | (z << 2); // SK:64
    ids[i] = (uint32_t)i;
}

// sorted points as float4 (xyz, original index bits) + the min/max box of every 1024 of them (SK:82-121)
__global__ void __launch_bounds__(TPB) gather_boxes_kernel(int P, const float *__restrict__ pts, const uint32_t *__restrict__ ids_sorted,
                                                            float4 *__restrict__ sp, Box *__restrict__ boxes)
{
    __shared__ float red[TPB / 64];
    Box me = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int q = 0; q < PPT; q++)
    {
        const int i = blockIdx.x * BOX + q * TPB + threadIdx.x;
        if (i < P)
        {
            const uint32_t id = ids_sorted[i];
            const float x = pts[3 * (size_t)id], y = pts[3 * (size_t)id + 1], z = pts[3 * (size_t)id + 2];
            sp[i] = make_float4(x, y, z, __uint_as_float(id));
            me.mnx = fminf(me.mnx, x); me.mny = fminf(me.mny, y); me.mnz = fminf(me.mnz, z);
            me.mxx = fmaxf(me.mxx, x); me.mxy = fmaxf(me.mxy, y); me.mxz = fmaxf(me.mxz, z);
        }
    }
    Box r;
    r.mnx = block_reduce(me.mnx, red, false); r.mny = block_reduce(me.mny, red, false); r.mnz = block_reduce(me.mnz, red, false);
    r.mxx = block_reduce(me.mxx, red, true); r.mxy = block_reduce(me.mxy, red, true); r.mxz = block_reduce(me.mxz, red, true);
    if (threadIdx.x == 0) boxes[blockIdx.x] = r;
}

// K = 3: mean squared distance to the 3 nearest others.  K = 1: nearest point of another group.
template <int K>
__global__ void __launch_bounds__(TPB) search_kernel(int P, int nboxes, int group, const float4 *__restrict__ sp,
                                                      const Box *__restrict__ boxes, float *__restrict__ out_mean,
                                                      uint32_t *__restrict__ out_nearest)
{
    __shared__ float4 cand[BOX];
    __shared__ uint32_t cand_group[K == 1 ? BOX : 1]; // group id of every staged candidate (one division per candidate per box)
    __shared__ float red[TPB / 64];
    const int mybox = blockIdx.x, tid = threadIdx.x;
    const Box bme = boxes[mybox];

    float px[PPT], py[PPT], pz[PPT], best[PPT][K];
    uint32_t pid[PPT], bestpos[PPT], bestid[PPT], mygroup[PPT];
    bool have[PPT];
#pragma unroll
    for (int q = 0; q < PPT; q++)
    {
        const int i = mybox * BOX + q * TPB + tid;
        have[q] = i < P;
        const float4 p = have[q] ? sp[i] : make_float4(0, 0, 0, 0);
        px[q] = p.x; py[q] = p.y; pz[q] = p.z; pid[q] = __float_as_uint(p.w);
        bestpos[q] = 0xFFFFFFFFu; bestid[q] = pid[q];
        mygroup[q] = pid[q] / (uint32_t)group;
#pragma unroll
        for (int k = 0; k < K; k++) best[q][k] = FLT_MAX;
    }

    auto scan_box = [&](int b) {
        const int n = min(BOX, P - b * BOX);
        __syncthreads(); // previous users of `cand` are done
        for (int i = tid; i < n; i += TPB)
        {
            const float4 c = sp[(size_t)b * BOX + i];
            cand[i] = c;
            if (K == 1) cand_group[i] = __float_as_uint(c.w) / (uint32_t)group;
        }
        __syncthreads();
        const Box bb = boxes[b];
#pragma unroll
        for (int q = 0; q < PPT; q++)
        {
            if (!have[q]) continue;
            if (dist_box_point(bb, px[q], py[q], pz[q]) > best[q][K - 1]) continue; // SK:174-175, 221-222
            const int self = (b == mybox) ? q * TPB + tid : -1;
            for (int i = 0; i < n; i++)
            {
                const float4 c = cand[i];
                const float d = dist2(px[q], py[q], pz[q], c.x, c.y, c.z);
                if (K == 3)
                {
                    if (i == self) continue; // SK:179-180: the point itself (by sorted position), duplicates count
                    float dd = d;            // SK:136-149
#pragma unroll
                    for (int k = 0; k < K; k++)
                        if (best[q][k] > dd) { const float t = best[q][k]; best[q][k] = dd; dd = t; }
                }
                else
                {
                    const uint32_t cid = __float_as_uint(c.w);
                    if (cand_group[i] == mygroup[q]) continue; // SK:226-227
                    const uint32_t pos = (uint32_t)(b * BOX + i);
                    if (d < best[q][0] || (d == best[q][0] && pos < bestpos[q])) // first in sorted order wins ties (SK:229)
                    {
                        best[q][0] = d; bestpos[q] = pos; bestid[q] = cid;
                    }
                }
            }
        }
    };

    scan_box(mybox);
    float worst = 0.0f; // this lane's largest current search radius
#pragma unroll
    for (int q = 0; q < PPT; q++)
        if (have[q]) worst = fmaxf(worst, best[q][K - 1]);
    float radius = block_reduce(worst, red, true);
    for (int b = 0; b < nboxes; b++)
    {
        if (b == mybox) continue;
        if (dist_box_box(bme, boxes[b]) > radius) continue; // workgroup-uniform: nobody here can gain from box b
        scan_box(b);
        worst = 0.0f;
#pragma unroll
        for (int q = 0; q < PPT; q++)
            if (have[q]) worst = fmaxf(worst, best[q][K - 1]);
        radius = block_reduce(worst, red, true);
    }
#pragma unroll
    for (int q = 0; q < PPT; q++)
    {
        if (!have[q]) continue;
        if (K == 3) out_mean[pid[q]] = (best[q][0] + best[q][1] + best[q][K - 1]) / 3.0f; // SK:186
        else out_nearest[pid[q]] = bestid[q];                                               // SK:234
    }
}

struct KnnCarve
{
    uint32_t *codes, *codes_sorted, *ids, *ids_sorted;
    float4 *sp;
    Box *boxes, *partial, *bbox;
    void *sort_temp;
    size_t sort_temp_bytes, bytes;
    int nboxes, npartial;
};

KnnCarve knn_carve(void *ws, int P)
{
    KnnCarve c;
    const size_t n = (size_t)(P > 0 ? P : 0);
    c.nboxes = (int)((n + BOX - 1) / BOX);
    c.npartial = 256;
    char *p = (char *)ts_align_up((size_t)ws);
    auto take = [&](size_t bytes) { char *q = p; p += ts_align_up(bytes); return q; };
    c.codes = (uint32_t *)take(n * 4); c.codes_sorted = (uint32_t *)take(n * 4);
    c.ids = (uint32_t *)take(n * 4); c.ids_sorted = (uint32_t *)take(n * 4);
    c.sp = (float4 *)take(n * 16);
    c.boxes = (Box *)take((size_t)c.nboxes * sizeof(Box));
    c.partial = (Box *)take((size_t)c.npartial * sizeof(Box));
    c.bbox = (Box *)take(sizeof(Box));
    c.sort_temp_bytes = ts_radix_scratch_bytes(n); // the hand-written radix sort of binning.hip
    c.sort_temp = take(c.sort_temp_bytes);
    c.bytes = (size_t)(p - (char *)ws);
    return c;
}

hipError_t knn_prepare(int P, const float *points, const KnnCarve &c, hipStream_t s)
{
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(c.npartial), dim3(TPB), 0, s, P, points, c.partial);
    hipLaunchKernelGGL(bbox_finish_kernel, dim3(1), dim3(64), 0, s, c.npartial, c.partial, c.bbox);
    hipLaunchKernelGGL(morton_kernel, dim3((P + TPB - 1) / TPB), dim3(TPB), 0, s, P, points, c.bbox, c.codes, c.ids);
    uint32_t *const k[2] = {c.codes, c.codes_sorted}, *const v[2] = {c.ids, c.ids_sorted};
    const int at = ts_radix_sort_pairs(k, v, (size_t)P, 30, c.sort_temp, s); // 30-bit Morton codes: four 8-bit passes, stable
    hipLaunchKernelGGL(gather_boxes_kernel, dim3(c.nboxes), dim3(TPB), 0, s, P, points, v[at], c.sp, c.boxes);
    return hipGetLastError();
}
} // namespace

size_t ts_knn_workspace_bytes(int P) { return knn_carve(nullptr, P).bytes + TS_ALIGN; }

hipError_t ts_knn_mean_dist3(int P, const float *points, float *mean_dist2, void *ws, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    const KnnCarve c = knn_carve(ws, P);
    hipError_t e = knn_prepare(P, points, c, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(search_kernel<3>, dim3(c.nboxes), dim3(TPB), 0, s, P, c.nboxes, 1, c.sp, c.boxes, mean_dist2, (uint32_t *)nullptr);
    return hipGetLastError();
}

hipError_t ts_knn_nearest_other(int P, int group, const float *points, uint32_t *nearest, void *ws, hipStream_t s)
{
    if (P <= 0) return hipSuccess;
    const KnnCarve c = knn_carve(ws, P);
    hipError_t e = knn_prepare(P, points, c, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(search_kernel<1>, dim3(c.nboxes), dim3(TPB), 0, s, P, c.nboxes, group, c.sp, c.boxes, (float *)nullptr, nearest);
    return hipGetLastError();
}
