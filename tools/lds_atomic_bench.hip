// Microbenchmark: throughput of LDS float atomics (ds_add_f32) under same-address sharing patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SHARE, int NVAL, bool USE_MAX>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    __shared__ float acc[4][64 * 20 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 20 + 64; i += 64) acc[wave][i] = 0.0f;
    __builtin_amdgcn_s_waitcnt(0);
    // lanes in groups of SHARE hit the same "entry"; entries are spread with stride 17 floats
    float x = (float)lane * 0.001f + 1.0f;
    int e = (lane / SHARE);
    for (int it = 0; it < iters; it++)
    {
        int ee = (e + it) & 63;
        float *p = &acc[wave][ee * 17];
#pragma unroll
        for (int v = 0; v < NVAL; v++)
        {
            if (USE_MAX) __hip_atomic_fetch_max(p + v, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(p + v, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        x = x * 1.0001f;
    }
    __syncthreads();
    float s = 0;
    for (int i = lane; i < 64 * 17; i += 64) s += acc[wave][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHARE, int NVAL, bool USE_MAX>
void run(const char *name, float *d, int blocks, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<SHARE, NVAL, USE_MAX>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<SHARE, NVAL, USE_MAX>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: blocks/256 blocks sequentially-ish (4 blocks/CU resident). wave-instr per CU = blocks/256*4 waves*iters*NVAL
    double winstr_per_cu = (double)blocks / 256.0 * 4.0 * iters * NVAL;
    double cyc = ms * 1e-3 * 2.4e9 / winstr_per_cu;
    printf("%-28s share=%2d nval=%2d  %.3f ms  ~%.1f cycles(@2.4GHz)/wave-instr/CU\n", name, SHARE, NVAL, ms, cyc);
}

int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    const int blocks = 2048, iters = 2000;
    run<1, 16, false>("add distinct", d, blocks, iters);
    run<2, 16, false>("add 2 lanes/addr", d, blocks, iters);
    run<4, 16, false>("add 4 lanes/addr", d, blocks, iters);
    run<8, 16, false>("add 8 lanes/addr", d, blocks, iters);
    run<16, 16, false>("add 16 lanes/addr", d, blocks, iters);
    run<64, 16, false>("add 64 lanes/addr", d, blocks, iters);
    run<1, 2, true>("max distinct", d, blocks, iters);
    run<16, 2, true>("max 16 lanes/addr", d, blocks, iters);
    run<64, 2, true>("max 64 lanes/addr", d, blocks, iters);
    return 0;
}
