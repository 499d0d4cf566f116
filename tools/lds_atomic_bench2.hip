// Microbenchmark: LDS atomic throughput on gfx950 by operation type (ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_max_i32), distinct
// addresses per lane and 2-4 lanes per address.  Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP, int SHARE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    __shared__ unsigned long long acc[4][64 * 3 + 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 3 + 64; i += 64) acc[wave][i] = 0;
    __syncthreads();
    float x = (float)lane * 0.001f + 1.0f;
    const int e = lane / SHARE;
    for (int it = 0; it < iters; it++)
    {
        const int ee = (e + it) & 63;
        unsigned long long *p = &acc[wave][ee * 3];
        if (OP == 0) __hip_atomic_fetch_add((float *)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 1) __hip_atomic_fetch_add((uint32_t *)p, (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 2) __hip_atomic_fetch_add(p, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 3) __hip_atomic_fetch_max((int *)p, it ^ lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 4) *(volatile float *)p = x; // plain ds_write_b32 for scale
        x = x * 1.0001f;
    }
    __syncthreads();
    float s = 0;
    for (int i = lane; i < 64 * 3; i += 64) s += (float)acc[wave][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int SHARE>
void run(const char *name, float *d)
{
    const int blocks = 2048, iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<OP, SHARE>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, SHARE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winstr_per_cu = (double)blocks / 256.0 * 4.0 * iters;
    printf("%-26s %d lane(s)/addr  %.3f ms  ~%.1f cycles(@2.4GHz)/wave-instr/CU\n", name, SHARE, ms, ms * 1e-3 * 2.4e9 / winstr_per_cu);
}

int main()
{
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    run<4, 1>("ds_write_b32 (scale)", d);
    run<0, 1>("ds_add_f32", d); run<0, 2>("ds_add_f32", d); run<0, 4>("ds_add_f32", d);
    run<1, 1>("ds_add_u32", d); run<1, 2>("ds_add_u32", d); run<1, 4>("ds_add_u32", d);
    run<2, 1>("ds_add_u64", d); run<2, 2>("ds_add_u64", d); run<2, 4>("ds_add_u64", d);
    run<3, 1>("ds_max_i32", d); run<3, 2>("ds_max_i32", d); run<3, 4>("ds_max_i32", d);
    return 0;
}
