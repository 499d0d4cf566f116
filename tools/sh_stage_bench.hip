// Microbenchmark: how fast can one lane per triangle consume a 192-byte SH row (48 floats) on gfx950?
//   A  single-wave workgroups, rows staged through LDS with scalar LDS writes, odd row stride 49   (= ts2d_stage.h today)
//   B  same, row stride 52, 128-bit LDS writes and reads
//   C  256-thread workgroups (4 waves share one launch), layout of B
//   D  no LDS: every lane reads its own row with 12 dwordx4 loads
//   E  like B but 2 row batches per workgroup, the second batch's global loads issued before the first is consumed
// Each lane reduces its row to one float (sum of squares) and writes 12 bytes.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ROW = 48;

__device__ __forceinline__ float consume(const float *r)
{
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < ROW; c++) s = fmaf(r[c], r[c], s);
    return s;
}

__global__ void __launch_bounds__(64) kA(const float *__restrict__ sh, float *__restrict__ out, int P)
{
    __shared__ float lds[64 * 49];
    const int lane = threadIdx.x, row0 = blockIdx.x * 64;
    const float *base = sh + (size_t)row0 * ROW;
#pragma unroll
    for (int it = 0; it < 12; it++)
    {
        const int i = (it * 64 + lane) * 4;
        const float4 q = *(const float4 *)(base + i);
        const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) lds[((i + k) / ROW) * 49 + ((i + k) % ROW)] = v[k];
    }
    __syncthreads();
    const float s = consume(lds + lane * 49);
    float *o = out + 3 * (size_t)(row0 + lane);
    o[0] = s; o[1] = s; o[2] = s;
}

template <int NT>
__global__ void __launch_bounds__(NT) kB(const float *__restrict__ sh, float *__restrict__ out, int P)
{
    __shared__ __attribute__((aligned(16))) float lds[NT * 52];
    const int t = threadIdx.x, row0 = blockIdx.x * NT;
    const float *base = sh + (size_t)row0 * ROW;
#pragma unroll
    for (int it = 0; it < 12; it++)
    {
        const int i = (it * NT + t) * 4; // ROW % 4 == 0: a float4 never straddles two rows
        *(float4 *)(lds + (i / ROW) * 52 + (i % ROW)) = *(const float4 *)(base + i);
    }
    __syncthreads();
    float r[ROW];
#pragma unroll
    for (int c = 0; c < ROW; c += 4) *(float4 *)(r + c) = *(const float4 *)(lds + t * 52 + c);
    const float s = consume(r);
    float *o = out + 3 * (size_t)(row0 + t);
    o[0] = s; o[1] = s; o[2] = s;
}

__global__ void __launch_bounds__(256) kD(const float *__restrict__ sh, float *__restrict__ out, int P)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const float4 *p = (const float4 *)(sh + (size_t)idx * ROW);
    float r[ROW];
#pragma unroll
    for (int c = 0; c < 12; c++) *(float4 *)(r + 4 * c) = p[c];
    const float s = consume(r);
    float *o = out + 3 * (size_t)idx;
    o[0] = s; o[1] = s; o[2] = s;
}

__global__ void __launch_bounds__(64) kE(const float *__restrict__ sh, float *__restrict__ out, int P)
{
    __shared__ __attribute__((aligned(16))) float lds[64 * 52];
    const int t = threadIdx.x;
    float4 q[12];
    const float *base = sh + (size_t)(blockIdx.x * 128) * ROW;
#pragma unroll
    for (int it = 0; it < 12; it++) q[it] = *(const float4 *)(base + (it * 64 + t) * 4);
    for (int half = 0; half < 2; half++)
    {
#pragma unroll
        for (int it = 0; it < 12; it++)
        {
            const int i = (it * 64 + t) * 4;
            *(float4 *)(lds + (i / ROW) * 52 + (i % ROW)) = q[it];
        }
        if (half == 0)
        {
            const float *nb = base + 64 * ROW;
#pragma unroll
            for (int it = 0; it < 12; it++) q[it] = *(const float4 *)(nb + (it * 64 + t) * 4);
        }
        __syncthreads();
        float r[ROW];
#pragma unroll
        for (int c = 0; c < ROW; c += 4) *(float4 *)(r + c) = *(const float4 *)(lds + t * 52 + c);
        const float s = consume(r);
        float *o = out + 3 * (size_t)(blockIdx.x * 128 + half * 64 + t);
        o[0] = s; o[1] = s; o[2] = s;
        __syncthreads();
    }
}

template <class F>
void run(const char *name, F launch, int P)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    printf("%-60s %7.1f us  %6.0f GB/s\n", name, ms * 1e3, (double)P * (ROW * 4 + 12) / ms / 1e6);
}

int main()
{
    const int P = 1 << 22;
    float *sh, *out;
    hipMalloc(&sh, (size_t)P * ROW * 4); hipMalloc(&out, (size_t)P * 12);
    hipMemset(sh, 0, (size_t)P * ROW * 4);
    run("A  64-thread WG, LDS stride 49, scalar LDS writes", [&] { hipLaunchKernelGGL(kA, dim3(P / 64), dim3(64), 0, 0, sh, out, P); }, P);
    run("B  64-thread WG, LDS stride 52, 128-bit LDS ops", [&] { hipLaunchKernelGGL(kB<64>, dim3(P / 64), dim3(64), 0, 0, sh, out, P); }, P);
    run("C  256-thread WG, LDS stride 52, 128-bit LDS ops", [&] { hipLaunchKernelGGL(kB<256>, dim3(P / 256), dim3(256), 0, 0, sh, out, P); }, P);
    run("D  no LDS, 12 dwordx4 loads per lane (192-byte lane stride)", [&] { hipLaunchKernelGGL(kD, dim3(P / 256), dim3(256), 0, 0, sh, out, P); }, P);
    run("E  64-thread WG, two batches, next batch prefetched", [&] { hipLaunchKernelGGL(kE, dim3(P / 128), dim3(64), 0, 0, sh, out, P); }, P);
    return 0;
}
