// Microbenchmark: cost of scattered global float atomics by scope on MI355X.  Each wave adds to 64 pseudo-random records
// (one float each, or a whole 64-byte record per 16 lanes) inside the slice of the array that belongs to ITS XCD
// (HW_REG_XCC_ID), with agent scope (sc1: serviced memory-side) or workgroup scope (serviced in the XCD's L2).
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics tools/atomic_scope_bench.hip -o tools/bin/atomic_scope_bench
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned xcc_id() { unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); return x & 7u; }

template <int SCOPE, int LINE>
__global__ void __launch_bounds__(256) k(float *a, unsigned per_xcd, int iters)
{
    const unsigned x = xcc_id();
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    float *base = a + (size_t)x * per_xcd * 16;
    for (int i = 0; i < iters; i++)
    {
        s = s * 1664525u + 1013904223u;
        unsigned rec;
        float *p;
        if (LINE) { rec = (unsigned)__shfl((int)(s >> 8), lane & 48) % per_xcd; p = base + (size_t)rec * 16 + (lane & 15); }
        else { rec = (s >> 8) % per_xcd; p = base + (size_t)rec * 16; }
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <typename K>
void run(const char *name, K kern, float *d, unsigned per_xcd, int iters)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipMemset(d, 0, (size_t)8 * per_xcd * 64);
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, per_xcd, 2);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(2048), dim3(256), 0, 0, d, per_xcd, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double lane_atomics = 2048.0 * 256 * iters;
    printf("%-34s per_xcd=%7u records: %.3f ms  %.1f G lane-atomics/s\n", name, per_xcd, ms, lane_atomics / ms * 1e-6);
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, (size_t)8 * 131072 * 64 + 64);
    for (unsigned per : {16384u, 131072u})
    {
        run("agent scope, 4 B scattered", k<0, 0>, d, per, 26);
        run("workgroup scope, 4 B scattered", k<1, 0>, d, per, 26);
        run("agent scope, 64 B line / 16 lanes", k<0, 1>, d, per, 26);
        run("workgroup scope, 64 B line / 16 lanes", k<1, 1>, d, per, 26);
    }
    // correctness of the L2-scope adds (each XCD only touches its own slice): total must equal the number of adds
    (void)hipMemset(d, 0, (size_t)8 * 16384 * 64);
    hipLaunchKernelGGL((k<1, 0>), dim3(2048), dim3(256), 0, 0, d, 16384u, 10);
    (void)hipDeviceSynchronize();
    static float h[8 * 16384 * 16];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double tot = 0;
    for (size_t i = 0; i < sizeof(h) / 4; i++) tot += h[i];
    printf("workgroup-scope sum check: %.0f (expected %.0f)\n", tot, 2048.0 * 256 * 10);
    return 0;
}
