// Microbenchmark (round 3): cost of ds_read_b128 / ds_read_b64 / ds_read_u16 when the 64 lanes of a wave read G distinct table rows
// (one per lane group of 64 / G lanes), for several row strides: the queue kernels read 8 rows per instruction where the lane-group
// kernels read 4.  Rows are picked pseudo-randomly among 32 slots per iteration.  Real shader cycles via s_memtime, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/lds_rows_bench.hip -o tools/bin/lds_rows_bench && tools/bin/lds_rows_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int GROUPS, int STRIDE, int MODE> // MODE 0: 4 x b128 per row  1: b64  2: u16
__global__ void __launch_bounds__(256) k_rows(float *out, long long *ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) char buf[4][33 * 256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 33 * 64; i += 64) ((float *)buf[wave])[i] = (float)i;
    const int grp = lane / (64 / GROUPS);
    unsigned state = 12345u + 977u * grp + blockIdx.x;
    float acc = 0.f;
    const unsigned base = (unsigned)(size_t)buf[wave];
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int r = 0; r < 8; r++)
        {
            state = state * 1664525u + 1013904223u;
            const unsigned slot = (state >> 16) & 31u;
            const unsigned a = base + slot * STRIDE + (MODE == 1 ? 8 * (lane & 7) : 0) + (MODE == 2 ? 2 * (lane & 7) : 0);
            if (MODE == 0)
            {
                float4 x0, x1, x2, x3;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(a) : "memory");
                acc += x0.x + x1.y + x2.z + x3.w;
            }
            else if (MODE == 1)
            {
                float2 x0;
                asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=&v"(x0) : "v"(a) : "memory");
                acc += x0.x + x0.y;
            }
            else
            {
                unsigned x0;
                asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)\n" : "=&v"(x0) : "v"(a) : "memory");
                acc += (float)x0;
            }
        }
    }
    const long long t1 = clock64();
    if (lane == 0) ticks[blockIdx.x * 4 + wave] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename K>
void run(const char *name, K kern, float *d, long long *dt, double inst_per_iter)
{
    const int blocks = 256 * 4, iters = 400; // 4 blocks of 33 KB per CU = 4 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dt, 4);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dt, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> t(blocks * 4);
    (void)hipMemcpy(t.data(), dt, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
    long long mx = 0;
    for (long long x : t) mx = x > mx ? x : mx;
    // 16 waves per CU share one LDS: LDS cycles per wave instruction = kernel cycles / (16 waves x instructions per wave)
    printf("%-34s %8.2f LDS cycles per wave instruction\n", name, (double)mx / (16.0 * iters * inst_per_iter));
    fflush(stdout);
}

int main()
{
    float *d; long long *dt;
    (void)hipMalloc(&d, 256 * 4 * 256 * 4);
    (void)hipMalloc(&dt, 256 * 4 * 4 * 8);
#define R(G, S, M, n) run("groups " #G " stride " #S " mode " #M, k_rows<G, S, M>, d, dt, n)
    R(1, 80, 0, 32.0); R(4, 80, 0, 32.0); R(8, 80, 0, 32.0); R(16, 80, 0, 32.0); R(64, 80, 0, 32.0);
    R(4, 208, 0, 32.0); R(8, 208, 0, 32.0); R(8, 144, 0, 32.0); R(8, 96, 0, 32.0); R(8, 112, 0, 32.0); R(8, 64, 0, 32.0); R(8, 128, 0, 32.0); R(8, 256, 0, 32.0);
    R(8, 176, 0, 32.0); R(8, 240, 0, 32.0); R(8, 272, 0, 32.0);
    R(4, 80, 1, 8.0); R(8, 80, 1, 8.0); R(8, 208, 1, 8.0); R(8, 144, 1, 8.0);
    R(4, 128, 2, 8.0); R(8, 128, 2, 8.0); R(8, 136, 2, 8.0);
    return 0;
}
