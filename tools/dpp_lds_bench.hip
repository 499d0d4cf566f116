// Does LDS allocation / LDS traffic slow down DPP and permlane VALU ops on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int LDS_FLOATS, int MODE, bool TRAFFIC>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    __shared__ float lds[LDS_FLOATS > 0 ? LDS_FLOATS : 1];
    float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.25f, d = a + 1.0f;
    if (LDS_FLOATS > 0) lds[threadIdx.x] = a;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++)
    {
        if (MODE == 0)
            asm volatile(REP32("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        else
            asm volatile(REP32("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (TRAFFIC) acc += lds[(threadIdx.x + i) & (LDS_FLOATS - 1)];
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + acc;
}

template <typename K>
void run(const char *name, K kern, float *d, int blocks_per_cu)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * blocks_per_cu, iters = 1000;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)blocks_per_cu * iters * 64.0;
    printf("%-40s %.3f ms  %.2f cycles/instr/SIMD (@2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run("dpp, no LDS, 4 blk/CU", k<0, 0, false>, d, 4);
    run("dpp, 40KB LDS alloc, 4 blk/CU", k<10240, 0, false>, d, 4);
    run("dpp, 40KB LDS + traffic", k<8192, 0, true>, d, 4);
    run("permlane, no LDS, 4 blk/CU", k<0, 1, false>, d, 4);
    run("permlane, 40KB LDS alloc", k<10240, 1, false>, d, 4);
    run("permlane, 40KB LDS + traffic", k<8192, 1, true>, d, 4);
    return 0;
}
