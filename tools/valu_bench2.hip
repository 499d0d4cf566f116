// Microbenchmark (round 2): issue cost of the instruction kinds a lane-group blend loop would use (gfx950):
// DPP flavours with partial bank masks, v_cndmask with VCC / SGPR-pair masks, VALU bit-walking ops, VALU under a
// half / quarter EXEC mask (does the SIMD-32 skip an all-zero pass?), and LDS reads with 1 / 4 distinct row addresses.
// Each kernel runs ITER x 32 copies of a 2-instruction body per wave; 8 waves per SIMD resident.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_bench2.hip -o /tmp/valu_bench2 && /tmp/valu_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

#define KERNEL(NAME, PRE, ASM)                                                                     \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters)                             \
    {                                                                                              \
        float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.25f, d = a + 1.0f;                        \
        float e = b, f = c, g = d, h = a;                                                          \
        unsigned long long m = 0x00ff00ff0f0f3355ull ^ blockIdx.x; int s = 3;                      \
        for (int i = 0; i < iters; i++)                                                            \
        {                                                                                          \
            asm volatile(PRE REP32(ASM) "s_mov_b64 exec, -1\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+s"(m), "+s"(s) : : "scc", "vcc"); \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (float)m + s;        \
    }

KERNEL(k_fma, "", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n")
KERNEL(k_add, "", "v_add_f32 %0, %1, %0\n v_add_f32 %3, %4, %3\n")
KERNEL(k_fma_half, "s_mov_b64 exec, 0xffffffff\n", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n")
KERNEL(k_fma_hi, "s_mov_b32 exec_lo, 0\n", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n")
KERNEL(k_fma_row, "s_mov_b64 exec, 0xffff\n", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %4, %5, %3\n")
KERNEL(k_exp_half, "s_mov_b64 exec, 0xffffffff\n", "v_exp_f32 %0, %1\n v_exp_f32 %3, %4\n")
KERNEL(k_cnd_vcc, "", "v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n")
KERNEL(k_cnd_sgpr, "", "v_cndmask_b32_e64 %0, %1, %2, %8\n v_cndmask_b32_e64 %3, %4, %5, %8\n")
KERNEL(k_cmp_vcc, "", "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %3, %4\n")
KERNEL(k_cmp_sgpr, "", "v_cmp_lt_f32_e64 %8, %0, %1\n v_cmp_lt_f32_e64 %8, %3, %4\n")
KERNEL(k_cmp_cnd, "", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %4, %5, vcc\n")
KERNEL(k_dpp_quad, "", "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_ror, "", "v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_bank, "", "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n")
KERNEL(k_dpp_mov, "", "v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_dpp_mirror, "", "v_add_f32_dpp %0, %1, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_mirror row_mask:0xf bank_mask:0xf\n")
KERNEL(k_ffbl, "", "v_ffbl_b32 %0, %1\n v_ffbl_b32 %3, %4\n")
KERNEL(k_andsub, "", "v_add_u32 %0, -1, %1\n v_and_b32 %3, %4, %3\n")
KERNEL(k_lshladd, "", "v_lshl_add_u32 %0, %1, 6, %2\n v_lshl_add_u32 %3, %4, 6, %5\n")
KERNEL(k_mad24, "", "v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %3, %4, %5, %3\n")
KERNEL(k_min3, "", "v_min3_f32 %0, %1, %2, %0\n v_min3_f32 %3, %4, %5, %3\n")
KERNEL(k_min, "", "v_min_f32 %0, %1, %0\n v_min_f32 %3, %4, %3\n")
KERNEL(k_max_i32, "", "v_max_i32 %0, %1, %0\n v_max_i32 %3, %4, %3\n")
KERNEL(k_swap32, "", "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %3, %4\n")
KERNEL(k_rcp, "", "v_rcp_f32 %0, %1\n v_rcp_f32 %3, %4\n")
KERNEL(k_fma_exp, "", "v_fma_f32 %0, %1, %2, %0\n v_exp_f32 %3, %4\n")
KERNEL(k_fma3_exp, "", "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %6, %1, %2, %6\n v_fma_f32 %7, %1, %2, %7\n v_exp_f32 %3, %4\n")
KERNEL(k_fma_dpp, "", "v_fma_f32 %0, %1, %2, %0\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n")
KERNEL(k_readlane, "", "v_readlane_b32 %9, %0, 5\n v_readlane_b32 %9, %3, 7\n")

// LDS: uniform-address (broadcast) ds_read_b128 vs four distinct row addresses (one per 16-lane group), rows 64 B or 80 B apart.
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[4][64 * 20];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 64 * 20; i += 64) buf[wave][i] = (float)i;
    float4 acc = make_float4(0, 0, 0, 0);
    int row = (MODE == 0) ? 0 : (lane >> 4) * 5; // wave-uniform vs one row per group
    const int stride = (MODE == 2) ? 20 : 16;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const float4 *p = (const float4 *)(buf[wave] + ((row + r * 3) & 63) * stride);
            float4 x0, x1, x2, x3;
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "v"((unsigned)(size_t)p) : "memory");
            acc.x += x0.x + x1.y; acc.y += x2.z + x3.w;
        }
        row = (row + 7) & 63;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}

template <typename K>
void run(const char *name, K kern, float *d, double instr_per_iter = 64.0)
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
    double instr_per_simd = 8.0 * iters * instr_per_iter;
    printf("%-14s %.3f ms  %.2f cycles/instr/SIMD (@2.4GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    fflush(stdout);
}

int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    run("v_fma_f32", k_fma, d); run("v_add_f32", k_add, d);
    run("fma exec=lo32", k_fma_half, d); run("fma exec=hi32", k_fma_hi, d); run("fma exec=row0", k_fma_row, d);
    run("exp exec=lo32", k_exp_half, d);
    run("cndmask vcc", k_cnd_vcc, d); run("cndmask sgpr", k_cnd_sgpr, d); run("v_cmp vcc", k_cmp_vcc, d);
    run("v_cmp sgpr", k_cmp_sgpr, d); run("cmp+cndmask", k_cmp_cnd, d);
    run("dpp quad_perm", k_dpp_quad, d); run("dpp row_ror", k_dpp_ror, d); run("dpp bank_mask", k_dpp_bank, d);
    run("dpp mov", k_dpp_mov, d); run("dpp mirror", k_dpp_mirror, d);
    run("v_ffbl_b32", k_ffbl, d); run("add/and u32", k_andsub, d); run("v_lshl_add", k_lshladd, d); run("v_mad_u32_u24", k_mad24, d);
    run("v_min3_f32", k_min3, d); run("v_min_f32", k_min, d); run("v_max_i32", k_max_i32, d);
    run("permlane32", k_swap32, d); run("v_rcp_f32", k_rcp, d);
    run("fma+exp", k_fma_exp, d); run("3fma+exp", k_fma3_exp, d, 128.0); run("fma+dpp", k_fma_dpp, d);
    run("v_readlane", k_readlane, d);
    // LDS: 16 x 4 ds_read_b128 per iteration
    run("lds b128 uni", k_lds<0>, d, 64.0); run("lds b128 4row", k_lds<1>, d, 64.0); run("lds b128 4r80", k_lds<2>, d, 64.0);
    return 0;
}
