// Calibration of rocprofv3's FETCH_SIZE for the blend kernels' access pattern (round 3): every lane gathers ONE 64-byte record
// (four dwordx4 loads) at a pseudo-random index of a table far larger than the L2s -- exactly what render_fwd / render_bwd do with the render
// records.  The kernel's byte count is known (records x 64 B, every record read once: a permutation), so FETCH_SIZE / known bytes is the factor
// to apply to the counter for this pattern (the guide's x2 holds for wide STREAMING reads).  A streaming kernel over the same bytes runs next to
// it for comparison.    rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/bin/gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) gather64(const float4 *__restrict__ rec, const unsigned *__restrict__ idx, float *out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 *p = rec + 4 * (size_t)idx[i];
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    out[i] = a.x + b.y + c.z + d.w;
}
__global__ void __launch_bounds__(256) stream64(const float4 *__restrict__ rec, float *out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 4 * n) return;
    const float4 a = rec[i];
    if (a.x == 123.456f) out[i & 1023] = a.y;
}
// Round 6: the WRITE side.  fill64: a streaming dwordx4 store of known size (what the zero / per-triangle kernels do); scatter64: every lane stores ONE
// 64-byte record at a pseudo-random index (the blend backward's row flush without the read-modify-write).   rocprofv3 --pmc WRITE_SIZE -- gather_calib
__global__ void __launch_bounds__(256) fill64(float4 *__restrict__ rec, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 4 * n) rec[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}
__global__ void __launch_bounds__(256) scatter64(float4 *__restrict__ rec, const unsigned *__restrict__ idx, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 *p = rec + 4 * (size_t)idx[i];
    const float4 v = make_float4((float)i, 1.0f, 2.0f, 3.0f);
    p[0] = v; p[1] = v; p[2] = v; p[3] = v;
}
__global__ void make_perm(unsigned *idx, unsigned n, unsigned mult)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) idx[i] = (unsigned)(((unsigned long long)i * mult) % n); // mult coprime to n: a permutation with a large stride
}

int main()
{
    const int n = 4 * 1000 * 1000 + 1; // records (256 MB): beyond the Infinity Cache
    float4 *rec; unsigned *idx; float *out;
    (void)hipMalloc(&rec, (size_t)n * 64);
    (void)hipMalloc(&idx, (size_t)n * 4);
    (void)hipMalloc(&out, (size_t)n * 4);
    (void)hipMemset(rec, 0, (size_t)n * 64);
    hipLaunchKernelGGL(make_perm, dim3((n + 255) / 256), dim3(256), 0, 0, idx, (unsigned)n, 2654435761u % (unsigned)n);
    for (int r = 0; r < 3; r++)
    {
        hipLaunchKernelGGL(gather64, dim3((n + 255) / 256), dim3(256), 0, 0, rec, idx, out, n);
        hipLaunchKernelGGL(stream64, dim3((4 * n + 255) / 256), dim3(256), 0, 0, rec, out, n);
        hipLaunchKernelGGL(fill64, dim3((4 * n + 255) / 256), dim3(256), 0, 0, rec, n);
        hipLaunchKernelGGL(scatter64, dim3((n + 255) / 256), dim3(256), 0, 0, rec, idx, n);
    }
    (void)hipDeviceSynchronize();
    printf("records %d, bytes per kernel %zu (+ %zu index bytes for gather64)\n", n, (size_t)n * 64, (size_t)n * 4);
    return 0;
}
