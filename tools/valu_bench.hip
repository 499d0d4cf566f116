// Microbenchmark: issue cost of the VALU instruction kinds used by the blend kernels (gfx950).
// Each kernel runs ITER x 32 independent copies of one instruction per wave; 8 waves/SIMD resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define KERNEL(NAME, ASM)                                                                          \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters)                             \
    {                                                                                              \
        float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.25f, d = a + 1.0f;                        \
        float e = b, f = c, g = d, h = a;                                                          \
        int s = 3;                                                                                 \
        for (int i = 0; i < iters; i++)                                                            \
        {                                                                                          \
            asm volatile(REP32(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(s) : : "scc", "vcc"); \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + s;                   \
    }

KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n")
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_pkfma(float *out, int iters)
{
    float2v a = {threadIdx.x * 0.5f, 1.0f}, b = {1.0001f, 0.5f}, c = {0.25f, 0.1f}, d = {2.0f, 3.0f};
    for (int i = 0; i < iters; i++)
    {
        asm volatile(REP32("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %3, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + d.x + d.y;
}
KERNEL(k_mul, "v_mul_f32 %0, %1, %0\n v_mul_f32 %3, %4, %3\n")
KERNEL(k_readlane, "v_readlane_b32 %8, %0, 5\n v_readlane_b32 %8, %3, 7\n")
KERNEL(k_dppadd, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_exp, "v_exp_f32 %0, %1\n v_exp_f32 %3, %4\n")
KERNEL(k_rcp, "v_rcp_f32 %0, %1\n v_rcp_f32 %3, %4\n")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %3, %4\n")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %3, %4\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n")
KERNEL(k_min3, "v_min3_f32 %0, %1, %2, %0\n v_min3_f32 %3, %4, %5, %3\n")
KERNEL(k_salu, "s_add_u32 %8, %8, 1\n s_add_u32 %8, %8, 3\n")
KERNEL(k_mix, "v_fma_f32 %0, %1, %2, %0\n s_add_u32 %8, %8, 1\n")

template <typename K>
void run(const char *name, K kern, float *d)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 8, iters = 2000; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = 8.0 * iters * 64.0; // waves per SIMD x iters x 64 instr
    fflush(stdout);
    printf("%-12s %.3f ms  %.2f cycles/instr/SIMD (@2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run("v_fma_f32", k_fma, d); run("v_pk_fma_f32", k_pkfma, d); run("v_mul_f32", k_mul, d);
    run("v_readlane", k_readlane, d); run("v_add_dpp", k_dppadd, d); run("v_exp_f32", k_exp, d);
    run("v_rcp_f32", k_rcp, d); run("permlane32", k_swap32, d); run("permlane16", k_swap16, d);
    run("v_cndmask", k_cndmask, d); run("v_min3_f32", k_min3, d); run("s_add_u32", k_salu, d);
    run("fma+salu", k_mix, d);
    fflush(stdout);
    return 0;
}
