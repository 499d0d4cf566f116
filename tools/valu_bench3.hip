// Microbenchmark (round 3): issue cost of the instruction kinds the blend loops use, in REAL shader cycles.
// valu_bench2.hip converted wall time at an assumed 2.4 GHz; this one brackets every wave's loop with s_memtime (clock64(): one
// tick = one shader cycle, MI355X_MICROARCH.md "Per-instruction cycle constants") and also reports the clock the chip actually
// sustained (ticks / wall time), so per-class costs and the kernels' GRBM_GUI_ACTIVE cycles share one clock.
// Grid = exactly the resident capacity (8 blocks x 4 waves per CU = 8 waves per SIMD), so every wave runs from kernel start to
// kernel end and  cycles per instruction per SIMD = mean wave ticks / (8 waves x instructions per wave).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_bench3.hip -o /tmp/valu_bench3 && /tmp/valu_bench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define KERNEL(NAME, PRE, ASM)                                                                     \
    __global__ void __launch_bounds__(256) NAME(float *out, long long *ticks, int iters)           \
    {                                                                                              \
        float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.25f, d = a + 1.0f;                        \
        float e = b, f = c, g = d, h = a;                                                          \
        unsigned long long m = 0x00ff00ff0f0f3355ull ^ blockIdx.x; int s = 3;                      \
        const long long t0 = clock64();                                                            \
        for (int i = 0; i < iters; i++)                                                            \
        {                                                                                          \
            asm volatile(PRE REP32(ASM) "s_mov_b64 exec, -1\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(m), "+s"(s) : : "scc", "vcc"); \
        }                                                                                          \
        const long long t1 = clock64();                                                            \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;         \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (float)m + s;        \
    }

KERNEL(k_fma, "", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n")
KERNEL(k_fma_ind, "", "v_fma_f32 %0, %1, %2, %6\n v_fma_f32 %3, %4, %5, %7\n")
KERNEL(k_mul, "", "v_mul_f32 %0, %1, %2\n v_mul_f32 %3, %4, %5\n")
KERNEL(k_add, "", "v_add_f32 %0, %1, %0\n v_add_f32 %3, %4, %3\n")
KERNEL(k_sub, "", "v_sub_f32 %0, %1, %2\n v_sub_f32 %3, %4, %5\n")
KERNEL(k_mov, "", "v_mov_b32 %0, %1\n v_mov_b32 %3, %4\n")
KERNEL(k_cnd_sgpr, "", "v_cndmask_b32_e64 %0, %1, %2, %8\n v_cndmask_b32_e64 %3, %4, %5, %8\n")
KERNEL(k_cmp_vcc, "", "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %3, %4\n")
KERNEL(k_cmp_sgpr, "", "v_cmp_lt_f32_e64 %8, %0, %1\n v_cmp_lt_f32_e64 %8, %3, %4\n")
KERNEL(k_cmp_cnd, "", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %4, %5, vcc\n")
KERNEL(k_dpp_quad, "", "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_ror, "", "v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_bank, "", "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n")
KERNEL(k_dpp_mov, "", "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_andsub, "", "v_add_u32 %0, -1, %1\n v_and_b32 %3, %4, %3\n")
KERNEL(k_lshladd, "", "v_lshl_add_u32 %0, %1, 6, %2\n v_lshl_add_u32 %3, %4, 6, %5\n")
KERNEL(k_lshl, "", "v_lshlrev_b32 %0, 3, %1\n v_lshrrev_b32 %3, 5, %4\n")
KERNEL(k_bfe, "", "v_bfe_u32 %0, %1, 8, 8\n v_bfe_i32 %3, %4, 0, 8\n")
KERNEL(k_mad24, "", "v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %3, %4, %5, %3\n")
KERNEL(k_bfi, "", "v_bfi_b32 %0, %1, %2, %0\n v_bfi_b32 %3, %4, %5, %3\n")
KERNEL(k_or3, "", "v_or3_b32 %0, %1, %2, %0\n v_and_or_b32 %3, %4, %5, %3\n")
KERNEL(k_xor, "", "v_xor_b32 %0, %1, %0\n v_or_b32 %3, %4, %3\n")
KERNEL(k_max_f, "", "v_max_f32 %0, %1, %0\n v_max_f32 %3, %4, %3\n")
KERNEL(k_fmac, "", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %4, %5\n")
KERNEL(k_cvt, "", "v_cvt_f32_u32 %0, %1\n v_cvt_u32_f32 %3, %4\n")
KERNEL(k_min3, "", "v_min3_f32 %0, %1, %2, %0\n v_min3_f32 %3, %4, %5, %3\n")
KERNEL(k_min, "", "v_min_f32 %0, %1, %0\n v_min_f32 %3, %4, %3\n")
KERNEL(k_swap32, "", "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %3, %4\n")
KERNEL(k_rcp, "", "v_rcp_f32 %0, %1\n v_rcp_f32 %3, %4\n")
KERNEL(k_exp, "", "v_exp_f32 %0, %1\n v_exp_f32 %3, %4\n")
KERNEL(k_fma3_exp, "", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %6, %1, %2, %6\n v_fma_f32 %7, %1, %2, %7\n v_exp_f32 %3, %4\n")
KERNEL(k_fma_dpp, "", "v_fma_f32 %0, %1, %2, %0\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_fma3_dpp, "", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %6, %1, %2, %6\n v_fma_f32 %7, %1, %2, %7\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_fma_cnd, "", "v_fma_f32 %0, %1, %2, %0\n v_cndmask_b32_e64 %3, %4, %5, %8\n")
KERNEL(k_readlane, "", "v_readlane_b32 %9, %0, 5\n v_readlane_b32 %9, %3, 7\n")
KERNEL(k_salu, "", "s_add_u32 %9, %9, 5\n s_lshl_b32 %9, %9, 1\n")
KERNEL(k_fma_salu, "", "v_fma_f32 %0, %1, %2, %0\n s_add_u32 %9, %9, 5\n")

// Packed fp32 (round 4; VERDICT r3 item 2: "is two pixels per lane a lever?"): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 work on 64-bit
// register pairs, two independent fp32 operations per lane and instruction.  If a wave instruction of these costs the same 2 cycles as the
// scalar form, a lane could carry two pixels (or two list entries) at half the issue cost of the full-rate part of the blend loops.
typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL_PK(NAME, ASM)                                                                       \
    __global__ void __launch_bounds__(256) NAME(float *out, long long *ticks, int iters)           \
    {                                                                                              \
        f2 a = {threadIdx.x * 0.5f, 1.5f}, b = {1.0001f, 0.9999f}, c = {0.25f, 0.5f}, d = a + 1.0f, e = b, f = c; \
        const long long t0 = clock64();                                                            \
        for (int i = 0; i < iters; i++) asm volatile(REP32(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)); \
        const long long t1 = clock64();                                                            \
        if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;         \
        out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + b.x + c.y + d.x + d.y + e.x + f.y;       \
    }
KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %3, %4, %5, %3\n")
KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %1, %2\n v_pk_mul_f32 %3, %4, %5\n")
KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %1, %0\n v_pk_add_f32 %3, %4, %3\n")

template <typename K>
void run(const char *name, K kern, float *d, long long *dt, int waves_per_simd, double instr_per_iter = 64.0)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd, iters = 2000;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dt, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dt, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<long long> t(blocks * 4);
    (void)hipMemcpy(t.data(), dt, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0; long long mx = 0;
    for (long long x : t) { sum += (double)x; if (x > mx) mx = x; }
    const double mean = sum / t.size();
    const double instr_per_simd = (double)waves_per_simd * iters * instr_per_iter;
    printf("%-14s w/simd %d  %.3f ms  mean wave ticks %.0f (max %lld)  -> %.2f cycles/instr/SIMD   clock %.2f GHz (ticks/wall)\n", name, waves_per_simd, ms,
           mean, mx, mean / instr_per_simd, mx / (ms * 1e-3) * 1e-9);
    fflush(stdout);
}

int main()
{
    float *d; long long *dt;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    (void)hipMalloc(&dt, 256 * 8 * 4 * 8);
#define RUN(n, k, ...) run(n, k, d, dt, 8, ##__VA_ARGS__)
    RUN("v_fma_f32", k_fma); RUN("v_fma indep", k_fma_ind); RUN("v_mul_f32", k_mul); RUN("v_add_f32", k_add); RUN("v_sub_f32", k_sub); RUN("v_mov_b32", k_mov);
    RUN("cndmask sgpr", k_cnd_sgpr); RUN("v_cmp vcc", k_cmp_vcc); RUN("v_cmp sgpr", k_cmp_sgpr); RUN("cmp+cndmask", k_cmp_cnd);
    RUN("dpp quad_perm", k_dpp_quad); RUN("dpp row_ror", k_dpp_ror); RUN("dpp bank_mask", k_dpp_bank); RUN("dpp mov", k_dpp_mov);
    RUN("add/and u32", k_andsub); RUN("v_lshl_add", k_lshladd); RUN("v_lshl/lshr", k_lshl); RUN("v_bfe", k_bfe); RUN("v_mad_u32_u24", k_mad24);
    RUN("v_bfi_b32", k_bfi); RUN("v_or3/and_or", k_or3); RUN("v_xor/or", k_xor); RUN("v_max_f32", k_max_f); RUN("v_fmac_f32", k_fmac); RUN("v_cvt", k_cvt);
    RUN("v_min3_f32", k_min3); RUN("v_min_f32", k_min);
    RUN("permlane32", k_swap32); RUN("v_rcp_f32", k_rcp); RUN("v_exp_f32", k_exp);
    RUN("3fma+exp", k_fma3_exp, 128.0); RUN("fma+dpp", k_fma_dpp); RUN("3fma+dpp", k_fma3_dpp, 128.0); RUN("fma+cndmask", k_fma_cnd);
    RUN("v_readlane", k_readlane); RUN("salu", k_salu); RUN("fma+salu", k_fma_salu);
    // packed fp32: two operations per lane and instruction
    RUN("v_pk_fma_f32", k_pk_fma); RUN("v_pk_mul_f32", k_pk_mul); RUN("v_pk_add_f32", k_pk_add);
    // fewer resident waves: does one wave reach the same rate?
    run("v_fma_f32", k_fma, d, dt, 1); run("v_fma_f32", k_fma, d, dt, 2); run("v_fma_f32", k_fma, d, dt, 4);
    run("dpp row_ror", k_dpp_ror, d, dt, 1); run("dpp row_ror", k_dpp_ror, d, dt, 2); run("dpp row_ror", k_dpp_ror, d, dt, 4);
    return 0;
}
