/*
 * ts2d_oracle.c -- CPU ORACLE for the 2D differentiable triangle rasterizer.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (triangle-splatting_amd/) never links, imports or calls it.
 *
 * What it is: a plain-C restatement of the algorithm of the reference's
 * submodules/diff-triangle-rasterization-2D ("R2D" below), written from the
 * reference's behaviour, function by function, each citing the R2D file:line it
 * follows.  All per-pair / per-triangle arithmetic is fp32 in the reference's
 * expression order; build with -ffp-contract=off so no FMA is formed.
 *
 * PARITY PINNING STATUS: pinned against the reference's own kernels.
 *   - The reference ships no tests, golden vectors or fixtures for this path
 *     (SURVEY.md section 4), and it is CUDA, so nothing can be pinned on a CPU.
 *     But the image carries the toolchain that installs CUDA extensions on ROCm
 *     (hipify-perl, hipcc, hipCUB / rocThrust, PyTorch): oracle/build_ref.py
 *     compiles the reference's three extensions for gfx950 from the sources
 *     where they lie into oracle/_ref/ (shared objects; no stand-in headers, nothing
 *     copied; the recipe's header lists every deviation), and
 *     tests/test_reference_gpu.py runs the reference's kernels on the MI355X
 *     against THIS oracle (2D and 3D variants, seven configurations each:
 *     integer state equal, images ~1e-5, gradients ~1e-5..1e-4) and against the
 *     HIP product path directly, up to the headline size.
 *   - Pinned on the CPU as well: the SH colour polynomial against the
 *     reference's Python `eval_sh` (src/diff_recon/utils/sh_utils.py:41-100)
 *     and the camera/matrix convention against src/diff_recon/utils/camera.py
 *     through committed fixtures (tests/golden/, generator script alongside).
 *   - The backward is additionally cross-checked against float64 torch autograd
 *     of an independent restatement of SURVEY Appendix A (tests/).
 *
 * Deliberate deviations from the reference (documented, order-only):
 *   - The reference accumulates per-triangle sums (contrib_sum, gradient
 *     scratch) with fp32 atomicAdd in an unspecified order.  The oracle adds
 *     the same fp32 per-pair terms into fp64 accumulators so that its result
 *     does not depend on thread order; it is rounded to fp32 at the end.
 *   - float->int conversions saturate like CUDA's cvt.rzi.s32.f32 (NaN -> 0)
 *     instead of x86's 0x80000000 indefinite value.
 *   - CUB's InclusiveSum / stable DeviceRadixSort::SortPairs (third-party, CUDA
 *     toolkit, unpinned; call sites R2D/src/rasterizer.cu:186,211) are restated
 *     by their published semantics: inclusive prefix sum; stable LSD radix sort
 *     of the low (32+bit) key bits.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TS_EPS ((float)(1e-8))     /* R2D/src/auxiliary.h:8 */
#define TS_BLOCK_X 16              /* R2D/src/config.h:4 */
#define TS_BLOCK_Y 16              /* R2D/src/config.h:5 */
#define TS_BLOCK_SIZE 256
#define TS_MAX_CHANNELS 3          /* R2D/src/config.h:3 */

/* SH constants, R2D/src/auxiliary.h:11-26 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { float x, y; } f2;
typedef struct { float x, y, z; } f3;
typedef struct { float x, y, z, w; } f4;

/* ---- small vector helpers (R2D/src/auxiliary.h:174-348), same expression order ---- */
static inline f2 f2_add(f2 a, f2 b) { f2 r = {a.x + b.x, a.y + b.y}; return r; }
static inline f2 f2_sub(f2 a, f2 b) { f2 r = {a.x - b.x, a.y - b.y}; return r; }
static inline f2 f2_mul(f2 a, f2 b) { f2 r = {a.x * b.x, a.y * b.y}; return r; }
static inline f2 f2_scale(f2 a, float s) { f2 r = {a.x * s, a.y * s}; return r; }      /* float2 * float */
static inline f2 f2_lscale(float s, f2 a) { f2 r = {s * a.x, s * a.y}; return r; }     /* float * float2 */
static inline f2 f2_addf(f2 a, float s) { f2 r = {a.x + s, a.y + s}; return r; }
static inline float f2_cross(f2 a, f2 b) { return a.x * b.y - a.y * b.x; }             /* auxiliary.h:174 */
static inline f2 f2_perp(f2 a) { f2 r = {a.y, -a.x}; return r; }                        /* cross(float2) auxiliary.h:184 */
static inline float f2_dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
static inline float f2_norm(f2 a) { return sqrtf(f2_dot(a, a)); }

static inline f3 f3_add(f3 a, f3 b) { f3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline f3 f3_sub(f3 a, f3 b) { f3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline f3 f3_lscale(float s, f3 a) { f3 r = {s * a.x, s * a.y, s * a.z}; return r; }
static inline f3 f3_scale(f3 a, float s) { f3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline f3 f3_div(f3 a, float s) { f3 r = {a.x / s, a.y / s, a.z / s}; return r; }
static inline float f3_dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float f3_norm(f3 a) { return sqrtf(f3_dot(a, a)); }
static inline f3 f3_cross(f3 a, f3 b)                                                    /* auxiliary.h:179 */
{
    f3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}

/* CUDA float->int (round toward zero, saturating, NaN->0) */
static inline int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* R2D/src/auxiliary.h:35-38 -- evaluated in double, returned as float */
static inline float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

/* R2D/src/auxiliary.h:40-48 */
static inline f3 xform_point_4x3(f3 p, const float *m)
{
    f3 r = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return r;
}
/* R2D/src/auxiliary.h:50-58 */
static inline f4 xform_point_4x4(f3 p, const float *m)
{
    f4 r = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
            m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
    return r;
}
/* R2D/src/auxiliary.h:60-67 */
static inline f3 xform_point_4x4_T(f4 p, const float *m)
{
    f3 r = {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * p.w,
            m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * p.w,
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * p.w};
    return r;
}
/* R2D/src/auxiliary.h:69-77 */
static inline f3 xform_vec_4x3(f3 p, const float *m)
{
    f3 r = {m[0] * p.x + m[4] * p.y + m[8] * p.z,
            m[1] * p.x + m[5] * p.y + m[9] * p.z,
            m[2] * p.x + m[6] * p.y + m[10] * p.z};
    return r;
}
/* R2D/src/auxiliary.h:79-87 */
static inline f3 xform_vec_4x3_T(f3 p, const float *m)
{
    f3 r = {m[0] * p.x + m[1] * p.y + m[2] * p.z,
            m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return r;
}
/* R2D/src/auxiliary.h:89-95 */
static inline f3 project_point(f3 p, const float *proj)
{
    f4 h = xform_point_4x4(p, proj);
    float w_inv = 1.0f / (fabsf(h.w) + TS_EPS);
    f3 r = {h.x * w_inv, h.y * w_inv, h.z * w_inv};
    return r;
}
/* R2D/src/auxiliary.h:97-118 -- first-order projection of a view-space vector at p_view */
static inline f2 project_vec_approx(f3 p, f3 v, float tx, float ty)
{
    f2 r = {(v.x - v.z * p.x / p.z) / (p.z * tx),
            (v.y - v.z * p.y / p.z) / (p.z * ty)};
    return r;
}
/* R2D/src/auxiliary.h:128-139 */
static inline f2 dnormvdv2(f2 v, f2 dv)
{
    float sum2 = v.x * v.x + v.y * v.y;
    float normv = sqrtf(sum2);
    float invsum32 = 1.0f / (normv * normv * normv);
    f2 r;
    r.x = ((sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y) * invsum32;
    return r;
}
/* R2D/src/auxiliary.h:141-152 */
static inline f3 dnormvdv3(f3 v, f3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float normv = sqrtf(sum2);
    float invsum32 = 1.0f / (normv * normv * normv);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* State kept between forward and backward.  Mirrors what the reference stores in its three   */
/* opaque buffers (R2D/src/param_struct.h:43-125).                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct ts2d_oracle_state
{
    int W, H, P, C, grid_x, grid_y, rich_info;
    int variant; /* 2 = R2D (screen-space barycentrics), 3 = R3D (view-space ray/plane barycentrics) */
    float tan_fovx, tan_fovy; /* R3D blend kernels need them (R3D/src/rasterizer.cu:232-233) */
    f3 *v1_view, *v2_view, *v3_view; /* R3D GeometryState, R3D/src/param_struct.h:46-48 */
    int64_t N; /* num_rendered */
    /* GeometryState, param_struct.h:46-58 */
    f2 *v1_2D, *v2_2D, *v3_2D;
    float *area2;
    f3 *normal_view, *v_depth;
    float *depth;
    float *rgb;           /* P*3 */
    uint8_t *clamped;     /* P*3 */
    uint32_t *point_offsets, *tiles_touched;
    uint32_t *rect_min, *rect_max; /* P*2 each */
    /* BinningState, param_struct.h:106-111 */
    uint64_t *keys_unsorted, *keys;
    uint32_t *vals_unsorted, *vals;
    /* ImageState, param_struct.h:88-90 */
    uint32_t *ranges;     /* T*2 */
    uint32_t *n_contrib;  /* W*H */
    float *final_T;       /* W*H */
} ts2d_oracle_state;

static void *zalloc(size_t n) { void *p = calloc(n ? n : 1, 1); return p; }

void ts2d_oracle_free(ts2d_oracle_state *s)
{
    if (!s) return;
    free(s->v1_2D); free(s->v2_2D); free(s->v3_2D); free(s->area2); free(s->normal_view);
    free(s->v_depth); free(s->depth); free(s->rgb); free(s->clamped); free(s->point_offsets);
    free(s->tiles_touched); free(s->rect_min); free(s->rect_max); free(s->keys_unsorted);
    free(s->keys); free(s->vals_unsorted); free(s->vals); free(s->ranges); free(s->n_contrib);
    free(s->final_T); free(s->v1_view); free(s->v2_view); free(s->v3_view);
    free(s);
}

/* R2D/src/rasterizer.cu:20-35 */
uint32_t ts2d_oracle_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1)
    {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* R2D/src/forward.cu:9-59 -- SH -> RGB at the direction centroid - campos */
static f3 rgb_from_sh(int idx, int deg, int max_coeffs, f3 pos, f3 campos, const float *shs, uint8_t *clamped)
{
    f3 dir = f3_sub(pos, campos);
    dir = f3_div(dir, f3_norm(dir));
    const f3 *sh = ((const f3 *)shs) + (size_t)idx * max_coeffs;
    f3 rgb = f3_lscale(SH_C0, sh[0]);
    if (deg > 0)
    {
        float x = dir.x, y = dir.y, z = dir.z;
        rgb = f3_sub(f3_add(f3_sub(rgb, f3_lscale(SH_C1 * y, sh[1])), f3_lscale(SH_C1 * z, sh[2])), f3_lscale(SH_C1 * x, sh[3]));
        if (deg > 1)
        {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            rgb = f3_add(rgb, f3_lscale(SH_C2[0] * xy, sh[4]));
            rgb = f3_add(rgb, f3_lscale(SH_C2[1] * yz, sh[5]));
            rgb = f3_add(rgb, f3_lscale(SH_C2[2] * (2.0f * zz - xx - yy), sh[6]));
            rgb = f3_add(rgb, f3_lscale(SH_C2[3] * xz, sh[7]));
            rgb = f3_add(rgb, f3_lscale(SH_C2[4] * (xx - yy), sh[8]));
            if (deg > 2)
            {
                rgb = f3_add(rgb, f3_lscale(SH_C3[0] * y * (3.0f * xx - yy), sh[9]));
                rgb = f3_add(rgb, f3_lscale(SH_C3[1] * xy * z, sh[10]));
                rgb = f3_add(rgb, f3_lscale(SH_C3[2] * y * (4.0f * zz - xx - yy), sh[11]));
                rgb = f3_add(rgb, f3_lscale(SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), sh[12]));
                rgb = f3_add(rgb, f3_lscale(SH_C3[4] * x * (4.0f * zz - xx - yy), sh[13]));
                rgb = f3_add(rgb, f3_lscale(SH_C3[5] * z * (xx - yy), sh[14]));
                rgb = f3_add(rgb, f3_lscale(SH_C3[6] * x * (xx - 3.0f * yy), sh[15]));
            }
        }
    }
    rgb.x += 0.5f; rgb.y += 0.5f; rgb.z += 0.5f;
    clamped[3 * idx + 0] = (rgb.x < 0);
    clamped[3 * idx + 1] = (rgb.y < 0);
    clamped[3 * idx + 2] = (rgb.z < 0);
    f3 r = {fmaxf(rgb.x, 0.0f), fmaxf(rgb.y, 0.0f), fmaxf(rgb.z, 0.0f)};
    return r;
}

/* R2D/src/forward.cu:61-193 -- per-triangle projection, culling, tile rect, colour */
static void preprocess_forward(ts2d_oracle_state *s, int D, int M, int use_shs, int back_culling,
                               float tan_fovx, float tan_fovy, const float *view, const float *proj,
                               const float *campos, const float *vertex, const float *shs, int *radii)
{
    const int W = s->W, H = s->H, P = s->P;
    const int gx = s->grid_x, gy = s->grid_y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++)
    {
        radii[idx] = 0;
        s->tiles_touched[idx] = 0;

        const f3 v1 = {vertex[9 * (size_t)idx], vertex[9 * (size_t)idx + 1], vertex[9 * (size_t)idx + 2]};
        const f3 v2 = {vertex[9 * (size_t)idx + 3], vertex[9 * (size_t)idx + 4], vertex[9 * (size_t)idx + 5]};
        const f3 v3 = {vertex[9 * (size_t)idx + 6], vertex[9 * (size_t)idx + 7], vertex[9 * (size_t)idx + 8]};
        const f3 center = f3_div(f3_add(f3_add(v1, v2), v3), 3.0f);
        const f3 center_proj = project_point(center, proj);

        if (center_proj.z <= 0) continue; /* near culling, forward.cu:98 */

        const f3 center_view = xform_point_4x3(center, view);
        const float limx = 1.3f * tan_fovx * center_view.z;
        const float limy = 1.3f * tan_fovy * center_view.z;
        const f3 cvc = {fminf(fmaxf(-limx, center_view.x), limx), fminf(fmaxf(-limy, center_view.y), limy), center_view.z};

        const f3 r1 = f3_sub(v1, center), r2 = f3_sub(v2, center), r3 = f3_sub(v3, center);
        const f3 r1_view = xform_vec_4x3(r1, view);
        const f3 r2_view = xform_vec_4x3(r2, view);
        if (f3_norm(f3_cross(r1_view, r2_view)) < TS_EPS) continue; /* forward.cu:113 */

        const f3 r3_view = xform_vec_4x3(r3, view);
        const f2 r1_proj = project_vec_approx(cvc, r1_view, tan_fovx, tan_fovy);
        const f2 r2_proj = project_vec_approx(cvc, r2_view, tan_fovx, tan_fovy);
        const f2 r3_proj = project_vec_approx(cvc, r3_view, tan_fovx, tan_fovy);
        const float n1 = f2_norm(r1_proj), n2 = f2_norm(r2_proj), n3 = f2_norm(r3_proj);
        if (n1 < TS_EPS || n2 < TS_EPS || n3 < TS_EPS) continue; /* forward.cu:124 */

        const f2 scaling = {0.5f * W, 0.5f * H};
        const float kernel_size = 0.5f;
        const f2 r1_2D = f2_mul(r1_proj, f2_addf(scaling, kernel_size / n1));
        const f2 r2_2D = f2_mul(r2_proj, f2_addf(scaling, kernel_size / n2));
        const f2 r3_2D = f2_mul(r3_proj, f2_addf(scaling, kernel_size / n3));
        const f2 center_2D = {ndc2pix(center_proj.x, W), ndc2pix(center_proj.y, H)};

        const f2 v1_2D = f2_add(center_2D, r1_2D);
        const f2 v2_2D = f2_add(center_2D, r2_2D);
        const f2 v3_2D = f2_add(center_2D, r3_2D);
        const float area2 = f2_cross(f2_sub(v2_2D, v1_2D), f2_sub(v3_2D, v1_2D));

        if (back_culling) { if (area2 >= -TS_EPS) continue; }      /* forward.cu:140-144 */
        else { if (fabsf(area2) < TS_EPS) continue; }              /* forward.cu:145-149 */

        const float dilation = 3.0f;
        const f2 d1 = f2_add(center_2D, f2_lscale(dilation, r1_2D));
        const f2 d2 = f2_add(center_2D, f2_lscale(dilation, r2_2D));
        const f2 d3 = f2_add(center_2D, f2_lscale(dilation, r3_2D));
        const f2 v_min = {fminf(fminf(d1.x, d2.x), d3.x), fminf(fminf(d1.y, d2.y), d3.y)};
        const f2 v_max = {fmaxf(fmaxf(d1.x, d2.x), d3.x), fmaxf(fmaxf(d1.y, d2.y), d3.y)};

        /* forward.cu:158-163 */
        const int rminx = imin(gx, imax(0, f2i_sat(v_min.x / TS_BLOCK_X)));
        const int rminy = imin(gy, imax(0, f2i_sat(v_min.y / TS_BLOCK_Y)));
        const int rmaxx = imin(gx, imax(0, f2i_sat((v_max.x + TS_BLOCK_X - 1) / TS_BLOCK_X)));
        const int rmaxy = imin(gy, imax(0, f2i_sat((v_max.y + TS_BLOCK_Y - 1) / TS_BLOCK_Y)));
        if (rmaxx <= rminx || rmaxy <= rminy) continue;

        if (use_shs)
        {
            const f3 cp = {campos[0], campos[1], campos[2]};
            f3 rgb = rgb_from_sh(idx, D, M, center, cp, shs, s->clamped);
            s->rgb[idx * 3 + 0] = rgb.x; s->rgb[idx * 3 + 1] = rgb.y; s->rgb[idx * 3 + 2] = rgb.z;
        }
        if (s->rich_info)
        {
            f3 n_view = f3_cross(r1_view, r2_view);
            n_view = f3_div(n_view, f3_norm(n_view));
            const f3 v_depth = {r1_view.z + center_view.z, r2_view.z + center_view.z, r3_view.z + center_view.z};
            s->normal_view[idx] = n_view;
            s->v_depth[idx] = v_depth;
        }
        s->v1_2D[idx] = v1_2D; s->v2_2D[idx] = v2_2D; s->v3_2D[idx] = v3_2D;
        s->area2[idx] = area2;
        s->depth[idx] = center_view.z;
        s->tiles_touched[idx] = (uint32_t)(rmaxx - rminx) * (uint32_t)(rmaxy - rminy);
        s->rect_min[2 * idx] = rminx; s->rect_min[2 * idx + 1] = rminy;
        s->rect_max[2 * idx] = rmaxx; s->rect_max[2 * idx + 1] = rmaxy;
        radii[idx] = f2i_sat(fmaxf(ceilf((v_max.x - v_min.x) * 0.5f), ceilf((v_max.y - v_min.y) * 0.5f)));
    }
}

/* Stable LSD radix sort of (u64 key, u32 value) on bits [0, end_bit) -- the published semantics of
 * cub::DeviceRadixSort::SortPairs as called at R2D/src/rasterizer.cu:211-218. */
static void radix_sort_pairs(uint64_t *k_in, uint32_t *v_in, uint64_t *k_out, uint32_t *v_out, int64_t n, int end_bit)
{
    uint64_t *ka = k_in, *kb = k_out;
    uint32_t *va = v_in, *vb = v_out;
    int passes = (end_bit + 7) / 8;
    uint64_t *ktmp = NULL; uint32_t *vtmp = NULL;
    /* keep the unsorted input intact: first pass reads input, alternates between out and tmp */
    if (passes == 0) { memcpy(k_out, k_in, n * sizeof(uint64_t)); memcpy(v_out, v_in, n * sizeof(uint32_t)); return; }
    ktmp = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    vtmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    /* choose ping-pong so that the final pass lands in k_out */
    uint64_t *dstk[2]; uint32_t *dstv[2];
    if (passes % 2 == 1) { dstk[0] = k_out; dstv[0] = v_out; dstk[1] = ktmp; dstv[1] = vtmp; }
    else { dstk[0] = ktmp; dstv[0] = vtmp; dstk[1] = k_out; dstv[1] = v_out; }
    for (int p = 0; p < passes; p++)
    {
        size_t count[257];
        memset(count, 0, sizeof(count));
        int shift = 8 * p;
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint64_t mask = ((uint64_t)1 << bits) - 1;
        for (int64_t i = 0; i < n; i++) count[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) count[d + 1] += count[d];
        kb = dstk[p & 1]; vb = dstv[p & 1];
        for (int64_t i = 0; i < n; i++)
        {
            size_t pos = count[(ka[i] >> shift) & mask]++;
            kb[pos] = ka[i]; vb[pos] = va[i];
        }
        ka = kb; va = vb;
    }
    free(ktmp); free(vtmp);
}

/* R2D/src/rasterizer.cu:186-231 host sequence: scan, key emission (:37-75), sort, tile ranges (:79-99) */
static void bin_and_sort(ts2d_oracle_state *s)
{
    const int P = s->P;
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += s->tiles_touched[i]; s->point_offsets[i] = run; } /* InclusiveSum */
    s->N = P > 0 ? (int64_t)(int32_t)s->point_offsets[P - 1] : 0;
    const int64_t N = s->N;
    s->keys_unsorted = (uint64_t *)zalloc(N * sizeof(uint64_t));
    s->keys = (uint64_t *)zalloc(N * sizeof(uint64_t));
    s->vals_unsorted = (uint32_t *)zalloc(N * sizeof(uint32_t));
    s->vals = (uint32_t *)zalloc(N * sizeof(uint32_t));

#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) /* duplicateWithKeys, rasterizer.cu:37-75 */
    {
        if (s->tiles_touched[idx] <= 0) continue;
        uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
        float d = s->depth[idx];
        uint32_t dbits; memcpy(&dbits, &d, 4);
        for (int y = (int)s->rect_min[2 * idx + 1]; y < (int)s->rect_max[2 * idx + 1]; y++)
            for (int x = (int)s->rect_min[2 * idx]; x < (int)s->rect_max[2 * idx]; x++)
            {
                uint64_t key = (uint64_t)(y * s->grid_x + x);
                key <<= 32;
                key |= dbits;
                s->keys_unsorted[off] = key;
                s->vals_unsorted[off] = (uint32_t)idx;
                off++;
            }
    }
    int bit = (int)ts2d_oracle_higher_msb((uint32_t)(s->grid_x * s->grid_y));
    radix_sort_pairs(s->keys_unsorted, s->vals_unsorted, s->keys, s->vals, N, 32 + bit);

    /* identifyTileRanges, rasterizer.cu:79-99 (ranges zeroed first, :223) */
    for (int64_t i = 0; i < N; i++)
    {
        uint32_t cur = (uint32_t)(s->keys[i] >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else
        {
            uint32_t prev = (uint32_t)(s->keys[i - 1] >> 32);
            if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == N - 1) s->ranges[2 * cur + 1] = (uint32_t)N;
    }
}

/* ---- R3D variant (submodules/diff-triangle-rasterization-3D, "R3D"): same host pipeline, different preprocess and
 * per-pixel mathematics.  R3D/src/auxiliary.h:35-43 */
static inline float proj_to_pix(float v, int S) { return (v + 1.0f) * S * 0.5f - 0.5f; }
static inline float pix_to_proj(float v, int S) { return (2.0f * v - S + 1.0f) / (float)(S); }

/* R3D/src/forward.cu:60-146 */
static void preprocess_forward_3d(ts2d_oracle_state *s, int D, int M, int use_shs, int back_culling, const float *view,
                                  const float *proj, const float *campos, const float *vertex, const float *shs, int *radii)
{
    const int W = s->W, H = s->H, P = s->P;
    const int gx = s->grid_x, gy = s->grid_y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++)
    {
        radii[idx] = 0;
        s->tiles_touched[idx] = 0;
        const f3 v1 = {vertex[9 * (size_t)idx], vertex[9 * (size_t)idx + 1], vertex[9 * (size_t)idx + 2]};
        const f3 v2 = {vertex[9 * (size_t)idx + 3], vertex[9 * (size_t)idx + 4], vertex[9 * (size_t)idx + 5]};
        const f3 v3 = {vertex[9 * (size_t)idx + 6], vertex[9 * (size_t)idx + 7], vertex[9 * (size_t)idx + 8]};
        const f3 v1_view = xform_point_4x3(v1, view), v2_view = xform_point_4x3(v2, view), v3_view = xform_point_4x3(v3, view);
        const f3 center_view = f3_div(f3_add(f3_add(v1_view, v2_view), v3_view), 3.0f);
        const f3 normal_view = f3_cross(f3_sub(v2_view, v1_view), f3_sub(v3_view, v1_view));
        if (f3_norm(normal_view) < TS_EPS) continue;           /* forward.cu:99 */
        if (back_culling && normal_view.z >= 0) continue;      /* forward.cu:101 */

        const float dilation = 3.0f;
        const f3 center = f3_div(f3_add(f3_add(v1, v2), v3), 3.0f);
        const f3 d1 = f3_add(center, f3_lscale(dilation, f3_sub(v1, center)));
        const f3 d2 = f3_add(center, f3_lscale(dilation, f3_sub(v2, center)));
        const f3 d3 = f3_add(center, f3_lscale(dilation, f3_sub(v3, center)));
        const f3 p1 = project_point(d1, proj), p2 = project_point(d2, proj), p3 = project_point(d3, proj);
        if (p1.z <= 0 || p2.z <= 0 || p3.z <= 0) continue;     /* near culling, forward.cu:114 */

        const f2 q1 = {proj_to_pix(p1.x, W), proj_to_pix(p1.y, H)};
        const f2 q2 = {proj_to_pix(p2.x, W), proj_to_pix(p2.y, H)};
        const f2 q3 = {proj_to_pix(p3.x, W), proj_to_pix(p3.y, H)};
        const f2 v_min = {fminf(fminf(q1.x, q2.x), q3.x), fminf(fminf(q1.y, q2.y), q3.y)};
        const f2 v_max = {fmaxf(fmaxf(q1.x, q2.x), q3.x), fmaxf(fmaxf(q1.y, q2.y), q3.y)};
        const int rminx = imin(gx, imax(0, f2i_sat(v_min.x / TS_BLOCK_X)));
        const int rminy = imin(gy, imax(0, f2i_sat(v_min.y / TS_BLOCK_Y)));
        const int rmaxx = imin(gx, imax(0, f2i_sat((v_max.x + TS_BLOCK_X - 1) / TS_BLOCK_X)));
        const int rmaxy = imin(gy, imax(0, f2i_sat((v_max.y + TS_BLOCK_Y - 1) / TS_BLOCK_Y)));
        if (rmaxx <= rminx || rmaxy <= rminy) continue;

        if (use_shs)
        {
            const f3 cp = {campos[0], campos[1], campos[2]};
            f3 rgb = rgb_from_sh(idx, D, M, center, cp, shs, s->clamped);
            s->rgb[idx * 3 + 0] = rgb.x; s->rgb[idx * 3 + 1] = rgb.y; s->rgb[idx * 3 + 2] = rgb.z;
        }
        s->v1_view[idx] = v1_view; s->v2_view[idx] = v2_view; s->v3_view[idx] = v3_view;
        s->normal_view[idx] = normal_view; /* NOT normalised in R3D */
        s->depth[idx] = center_view.z;
        s->tiles_touched[idx] = (uint32_t)(rmaxx - rminx) * (uint32_t)(rmaxy - rminy);
        s->rect_min[2 * idx] = rminx; s->rect_min[2 * idx + 1] = rminy;
        s->rect_max[2 * idx] = rmaxx; s->rect_max[2 * idx + 1] = rmaxy;
        radii[idx] = f2i_sat(fmaxf(ceilf((v_max.x - v_min.x) * 0.5f), ceilf((v_max.y - v_min.y) * 0.5f)));
    }
}

static inline void atomic_add_d(double *p, double v)
{
#pragma omp atomic
    *p += v;
}
static inline void atomic_max_f(float *p, float v)
{
    /* R2D/src/auxiliary.h:350-356 (atomicMaxFloat).  contrib = alpha*T is always > 0 and the
     * target starts at 0, so the non-negative branch (integer max on the bit pattern) is the only
     * one reachable; max is order independent. */
    int32_t vi; memcpy(&vi, &v, 4);
    int32_t *ip = (int32_t *)p;
    int32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED);
    while (old < vi && !__atomic_compare_exchange_n(ip, &old, vi, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

/* R2D/src/forward.cu:198-355 -- per-pixel front-to-back blend of the tile's depth-sorted list */
static void render_forward(ts2d_oracle_state *s, float gamma, const float *feature, const float *opacity,
                           float background_depth, const float *background, float *out_feature,
                           float *out_depth, float *out_normal, double *contrib_sum_d, float *contrib_max)
{
    const int W = s->W, H = s->H, C = s->C, rich = s->rich_info;
    const int ntiles = s->grid_x * s->grid_y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++)
    {
        const int tx = tile % s->grid_x, ty = tile / s->grid_x;
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < TS_BLOCK_Y; ly++)
            for (int lx = 0; lx < TS_BLOCK_X; lx++)
            {
                const uint32_t px = tx * TS_BLOCK_X + lx, py = ty * TS_BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue; /* "inside", forward.cu:232 */
                const uint32_t pix_id = W * py + px;
                const f2 pixf = {(float)px, (float)py};
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float accum_feature[TS_MAX_CHANNELS] = {0, 0, 0};
                f3 accum_normal = {0, 0, 0};
                float accum_depth = 0.0f;
                for (uint32_t k = r0; k < r1; k++)
                {
                    contributor++;
                    last_contributor = contributor;
                    const uint32_t id = s->vals[k];
                    const float area2 = s->area2[id];
                    const f2 p_v1 = f2_sub(s->v1_2D[id], pixf);
                    const f2 p_v2 = f2_sub(s->v2_2D[id], pixf);
                    const f2 p_v3 = f2_sub(s->v3_2D[id], pixf);
                    const float a1 = f2_cross(p_v2, p_v3) / area2;
                    const float a2 = f2_cross(p_v3, p_v1) / area2;
                    const float a3 = 1.0f - a1 - a2;
                    const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
                    if (ecc < 0.0f || ecc > 10.0f) continue;
                    const float power = -0.5f * powf(ecc, 2.0f * gamma);
                    const float alpha = fminf(0.99f, opacity[id] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float contrib = alpha * T;
                    for (int ch = 0; ch < C; ch++) accum_feature[ch] += feature[id * C + ch] * contrib;
                    if (rich)
                    {
                        atomic_add_d(&contrib_sum_d[id], (double)contrib);
                        atomic_max_f(&contrib_max[id], contrib);
                        const f3 nv = s->normal_view[id];
                        accum_normal.x += nv.x * contrib; accum_normal.y += nv.y * contrib; accum_normal.z += nv.z * contrib;
                        const f3 vd = s->v_depth[id];
                        const float d = vd.x * a1 + vd.y * a2 + vd.z * a3;
                        accum_depth += d * contrib;
                    }
                    T *= (1.0f - alpha);
                    if (T <= 0.0001f) break; /* "done", forward.cu:333-334: stop AFTER including this one */
                }
                s->final_T[pix_id] = T;
                s->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < C; ch++) out_feature[(size_t)ch * H * W + pix_id] = accum_feature[ch] + T * background[ch];
                if (rich)
                {
                    out_depth[pix_id] = accum_depth + T * background_depth;
                    out_normal[pix_id] = accum_normal.x;
                    out_normal[(size_t)H * W + pix_id] = accum_normal.y;
                    out_normal[2 * (size_t)H * W + pix_id] = accum_normal.z;
                }
            }
    }
}

static void rgb_from_sh_backward(int idx, int deg, int max_coeffs, f3 pos, f3 campos, const float *shs,
                                 const uint8_t *clamped, const f3 *dL_dfeature, f3 *dL_dshs, f3 *dL_dpos);

/* R3D/src/forward.cu:151-306 -- per-pixel ray / triangle-plane intersection in view space */
static void render_forward_3d(ts2d_oracle_state *s, float gamma, const float *feature, const float *opacity,
                              float background_depth, const float *background, float *out_feature, float *out_depth,
                              float *out_normal, double *contrib_sum_d, float *contrib_max)
{
    const int W = s->W, H = s->H, C = s->C, rich = s->rich_info;
    const int ntiles = s->grid_x * s->grid_y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++)
    {
        const int tx = tile % s->grid_x, ty = tile / s->grid_x;
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < TS_BLOCK_Y; ly++)
            for (int lx = 0; lx < TS_BLOCK_X; lx++)
            {
                const uint32_t px = tx * TS_BLOCK_X + lx, py = ty * TS_BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const uint32_t pix_id = W * py + px;
                const f3 p_ray = {s->tan_fovx * pix_to_proj((float)px, W), s->tan_fovy * pix_to_proj((float)py, H), 1.0f};
                float T = 1.0f;
                uint32_t contributor = 0, last_contributor = 0;
                float accum_feature[TS_MAX_CHANNELS] = {0, 0, 0};
                f3 accum_normal = {0, 0, 0};
                float accum_depth = 0.0f;
                for (uint32_t k = r0; k < r1; k++)
                {
                    contributor++;
                    last_contributor = contributor;
                    const uint32_t id = s->vals[k];
                    const f3 v1 = s->v1_view[id], v2 = s->v2_view[id], v3 = s->v3_view[id], n = s->normal_view[id];
                    const float p_ray_dot_n = f3_dot(p_ray, n);
                    if (fabsf(p_ray_dot_n) < TS_EPS) continue;
                    const float depth = f3_dot(v1, n) / p_ray_dot_n;
                    const f3 p_view = f3_lscale(depth, p_ray);
                    const f3 p_v1 = f3_sub(v1, p_view), p_v2 = f3_sub(v2, p_view), p_v3 = f3_sub(v3, p_view);
                    const float inv_n_dot_n = 1.0f / f3_dot(n, n);
                    const float a1 = f3_dot(f3_cross(p_v2, p_v3), n) * inv_n_dot_n;
                    const float a2 = f3_dot(f3_cross(p_v3, p_v1), n) * inv_n_dot_n;
                    const float a3 = 1.0f - a1 - a2;
                    const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
                    if (ecc < 0.0f || ecc > 10.0f) continue;
                    const float power = -0.5f * powf(ecc, 2.0f * gamma);
                    const float alpha = fminf(0.99f, opacity[id] * expf(power));
                    if (alpha < 1.0f / 255.0f) continue;
                    const float contrib = alpha * T;
                    T *= (1.0f - alpha);
                    for (int ch = 0; ch < C; ch++) accum_feature[ch] += feature[id * C + ch] * contrib;
                    if (rich)
                    {
                        atomic_add_d(&contrib_sum_d[id], (double)contrib);
                        atomic_max_f(&contrib_max[id], contrib);
                        accum_normal.x += n.x * contrib; accum_normal.y += n.y * contrib; accum_normal.z += n.z * contrib;
                        accum_depth += depth * contrib;
                    }
                    if (T <= 0.0001f) break;
                }
                s->final_T[pix_id] = T;
                s->n_contrib[pix_id] = last_contributor;
                for (int ch = 0; ch < C; ch++) out_feature[(size_t)ch * H * W + pix_id] = accum_feature[ch] + T * background[ch];
                if (rich)
                {
                    out_depth[pix_id] = accum_depth + T * background_depth;
                    out_normal[pix_id] = accum_normal.x;
                    out_normal[(size_t)H * W + pix_id] = accum_normal.y;
                    out_normal[2 * (size_t)H * W + pix_id] = accum_normal.z;
                }
            }
    }
}

/* R3D/src/backward.cu:216-454.  g_v: P*9 (v1,v2,v3 view-space), g_normal: P*3, g_feature: P*C, g_opacity: P.
 * Keeps the reference's quirk that the skip test here is on G, not on alpha (backward.cu:351 vs forward.cu:265). */
static void render_backward_3d(const ts2d_oracle_state *s, float gamma, const float *feature, const float *opacity,
                               float background_depth, const float *background, const float *dL_dout_feature,
                               const float *dL_dout_depth, const float *dL_dout_normal, double *g_v, double *g_normal,
                               double *g_feature, double *g_opacity)
{
    const int W = s->W, H = s->H, C = s->C, rich = s->rich_info;
    const int ntiles = s->grid_x * s->grid_y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++)
    {
        const int tx = tile % s->grid_x, ty = tile / s->grid_x;
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < TS_BLOCK_Y; ly++)
            for (int lx = 0; lx < TS_BLOCK_X; lx++)
            {
                const uint32_t px = tx * TS_BLOCK_X + lx, py = ty * TS_BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const uint32_t pix_id = W * py + px;
                const f3 p_ray = {s->tan_fovx * pix_to_proj((float)px, W), s->tan_fovy * pix_to_proj((float)py, H), 1.0f};
                float T = s->final_T[pix_id];
                const uint32_t last_contributor = s->n_contrib[pix_id];
                uint32_t contributor = r1 - r0;
                float accum_feature[TS_MAX_CHANNELS] = {0, 0, 0};
                f3 accum_normal = {0, 0, 0};
                float accum_depth = background_depth;
                float dL_dfeature_pixel[TS_MAX_CHANNELS] = {0, 0, 0};
                f3 dL_dnormal_pixel = {0, 0, 0};
                float dL_ddepth_pixel = 0;
                for (int i = 0; i < C; i++)
                {
                    accum_feature[i] = background[i];
                    dL_dfeature_pixel[i] = dL_dout_feature[(size_t)i * H * W + pix_id];
                }
                if (rich)
                {
                    dL_dnormal_pixel.x = dL_dout_normal[pix_id];
                    dL_dnormal_pixel.y = dL_dout_normal[(size_t)W * H + pix_id];
                    dL_dnormal_pixel.z = dL_dout_normal[2 * (size_t)W * H + pix_id];
                    dL_ddepth_pixel = dL_dout_depth[pix_id];
                }
                for (uint32_t kk = r1; kk > r0; kk--)
                {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t id = s->vals[kk - 1];
                    const f3 v1 = s->v1_view[id], v2 = s->v2_view[id], v3 = s->v3_view[id], n = s->normal_view[id];
                    const float p_ray_dot_n = f3_dot(p_ray, n);
                    if (fabsf(p_ray_dot_n) < TS_EPS) continue;
                    const float inv_p_ray_dot_n = 1.0f / p_ray_dot_n;
                    const float depth = f3_dot(v1, n) * inv_p_ray_dot_n;
                    const f3 p_view = f3_lscale(depth, p_ray);
                    const f3 p_v1 = f3_sub(v1, p_view), p_v2 = f3_sub(v2, p_view), p_v3 = f3_sub(v3, p_view);
                    const float inv_n_dot_n = 1.0f / f3_dot(n, n);
                    const float a1 = f3_dot(f3_cross(p_v2, p_v3), n) * inv_n_dot_n;
                    const float a2 = f3_dot(f3_cross(p_v3, p_v1), n) * inv_n_dot_n;
                    const float a3 = 1.0f - a1 - a2;
                    const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
                    if (ecc < 0.0f || ecc > 10.0f) continue;
                    const float power = -0.5f * powf(ecc, 2.0f * gamma);
                    const float op = opacity[id];
                    const float G = expf(power);
                    const float alpha = fminf(0.99f, op * G);
                    if (G < 1.0f / 255.0f) continue; /* sic: G, backward.cu:351 */

                    T /= (1.0f - alpha);
                    const float contrib = alpha * T;
                    float dL_dcontrib = 0.0f;
                    f3 dL_dnormal = {0, 0, 0};
                    float dL_ddepth = 0.0f;
                    for (int ch = 0; ch < C; ch++)
                    {
                        atomic_add_d(&g_feature[(size_t)id * C + ch], (double)(dL_dfeature_pixel[ch] * contrib));
                        const float feat = feature[id * C + ch];
                        dL_dcontrib += dL_dfeature_pixel[ch] * (feat - accum_feature[ch]);
                        accum_feature[ch] = alpha * feat + (1.0f - alpha) * accum_feature[ch];
                    }
                    if (rich)
                    {
                        dL_dnormal = f3_add(dL_dnormal, f3_scale(dL_dnormal_pixel, contrib));
                        dL_dcontrib += f3_dot(dL_dnormal_pixel, f3_sub(n, accum_normal));
                        accum_normal = f3_add(f3_lscale(alpha, n), f3_lscale(1.0f - alpha, accum_normal));
                        dL_ddepth += dL_ddepth_pixel * contrib;
                        dL_dcontrib += dL_ddepth_pixel * (depth - accum_depth);
                        accum_depth = alpha * depth + (1.0f - alpha) * accum_depth;
                    }
                    const float dL_dalpha = dL_dcontrib * T;
                    const float dL_dpower = (op * G < 0.99f) ? (dL_dalpha * alpha) : 0.0f;
                    const float dL_decc = dL_dpower * 2 * gamma * power / (ecc + TS_EPS);
                    f3 decc_da = {0, 0, 0};
                    if (a1 <= a2 && a1 <= a3) decc_da.x = -3.0f;
                    else if (a2 <= a1 && a2 <= a3) decc_da.y = -3.0f;
                    else decc_da.z = -3.0f;
                    const f3 dL_da = f3_lscale(dL_decc, decc_da);

                    const f3 zero = {0, 0, 0};
                    const f3 da1_dv1 = zero;
                    const f3 da1_dv2 = f3_scale(f3_cross(p_v3, n), inv_n_dot_n);
                    const f3 da1_dv3 = f3_scale(f3_cross(n, p_v2), inv_n_dot_n);
                    const f3 da1_dn = f3_scale(f3_sub(f3_cross(p_v2, p_v3), f3_lscale(2.0f * a1, n)), inv_n_dot_n);
                    const float da1_dd = f3_dot(n, f3_cross(f3_sub(v3, v2), p_ray)) * inv_n_dot_n;
                    const f3 da2_dv1 = f3_scale(f3_cross(n, p_v3), inv_n_dot_n);
                    const f3 da2_dv2 = zero;
                    const f3 da2_dv3 = f3_scale(f3_cross(p_v1, n), inv_n_dot_n);
                    const f3 da2_dn = f3_scale(f3_sub(f3_cross(p_v3, p_v1), f3_lscale(2.0f * a2, n)), inv_n_dot_n);
                    const float da2_dd = f3_dot(n, f3_cross(f3_sub(v1, v3), p_ray)) * inv_n_dot_n;
                    const f3 neg1 = {-da1_dv1.x, -da1_dv1.y, -da1_dv1.z}, neg2 = {-da1_dv2.x, -da1_dv2.y, -da1_dv2.z};
                    const f3 neg3 = {-da1_dv3.x, -da1_dv3.y, -da1_dv3.z}, negn = {-da1_dn.x, -da1_dn.y, -da1_dn.z};
                    const f3 da3_dv1 = f3_sub(neg1, da2_dv1), da3_dv2 = f3_sub(neg2, da2_dv2), da3_dv3 = f3_sub(neg3, da2_dv3);
                    const f3 da3_dn = f3_sub(negn, da2_dn);
                    const float da3_dd = -da1_dd - da2_dd;

                    dL_ddepth += dL_da.x * da1_dd + dL_da.y * da2_dd + dL_da.z * da3_dd;
                    const f3 ddepth_dv1 = f3_scale(n, inv_p_ray_dot_n);
                    const f3 ddepth_dn = f3_scale(f3_sub(v1, f3_lscale(depth, p_ray)), inv_p_ray_dot_n);

                    const f3 gv1 = f3_add(f3_add(f3_add(f3_lscale(dL_da.x, da1_dv1), f3_lscale(dL_da.y, da2_dv1)), f3_lscale(dL_da.z, da3_dv1)),
                                          f3_lscale(dL_ddepth, ddepth_dv1));
                    const f3 gv2 = f3_add(f3_add(f3_lscale(dL_da.x, da1_dv2), f3_lscale(dL_da.y, da2_dv2)), f3_lscale(dL_da.z, da3_dv2));
                    const f3 gv3 = f3_add(f3_add(f3_lscale(dL_da.x, da1_dv3), f3_lscale(dL_da.y, da2_dv3)), f3_lscale(dL_da.z, da3_dv3));
                    dL_dnormal = f3_add(dL_dnormal, f3_add(f3_add(f3_add(f3_lscale(dL_da.x, da1_dn), f3_lscale(dL_da.y, da2_dn)),
                                                                  f3_lscale(dL_da.z, da3_dn)), f3_lscale(dL_ddepth, ddepth_dn)));
                    double *gv = g_v + 9 * (size_t)id;
                    atomic_add_d(gv + 0, gv1.x); atomic_add_d(gv + 1, gv1.y); atomic_add_d(gv + 2, gv1.z);
                    atomic_add_d(gv + 3, gv2.x); atomic_add_d(gv + 4, gv2.y); atomic_add_d(gv + 5, gv2.z);
                    atomic_add_d(gv + 6, gv3.x); atomic_add_d(gv + 7, gv3.y); atomic_add_d(gv + 8, gv3.z);
                    atomic_add_d(g_normal + 3 * (size_t)id + 0, dL_dnormal.x);
                    atomic_add_d(g_normal + 3 * (size_t)id + 1, dL_dnormal.y);
                    atomic_add_d(g_normal + 3 * (size_t)id + 2, dL_dnormal.z);
                    atomic_add_d(&g_opacity[id], (double)(dL_dalpha * G));
                }
            }
    }
}

/* R3D/src/backward.cu:144-214 */
static void preprocess_backward_3d(const ts2d_oracle_state *s, int D, int M, int use_shs, const float *view, const float *campos,
                                   const float *vertex, const float *shs, const int *radii, const float *g_v,
                                   const float *g_normal, const float *dL_dfeature, float *dL_dvertex, float *dL_dcenter2D,
                                   float *dL_dshs)
{
    const int P = s->P;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++)
    {
        if (radii[idx] <= 0) continue;
        const f3 v1_view = s->v1_view[idx], v2_view = s->v2_view[idx], v3_view = s->v3_view[idx];
        f3 g1 = {g_v[9 * (size_t)idx + 0], g_v[9 * (size_t)idx + 1], g_v[9 * (size_t)idx + 2]};
        f3 g2 = {g_v[9 * (size_t)idx + 3], g_v[9 * (size_t)idx + 4], g_v[9 * (size_t)idx + 5]};
        f3 g3 = {g_v[9 * (size_t)idx + 6], g_v[9 * (size_t)idx + 7], g_v[9 * (size_t)idx + 8]};
        const f3 gn = {g_normal[3 * (size_t)idx], g_normal[3 * (size_t)idx + 1], g_normal[3 * (size_t)idx + 2]};
        g1 = f3_add(g1, f3_cross(f3_sub(v2_view, v3_view), gn));
        g2 = f3_add(g2, f3_cross(f3_sub(v3_view, v1_view), gn));
        g3 = f3_add(g3, f3_cross(f3_sub(v1_view, v2_view), gn));
        f3 dL_dv1 = xform_vec_4x3_T(g1, view), dL_dv2 = xform_vec_4x3_T(g2, view), dL_dv3 = xform_vec_4x3_T(g3, view);
        if (use_shs)
        {
            const f3 v1 = {vertex[9 * (size_t)idx], vertex[9 * (size_t)idx + 1], vertex[9 * (size_t)idx + 2]};
            const f3 v2 = {vertex[9 * (size_t)idx + 3], vertex[9 * (size_t)idx + 4], vertex[9 * (size_t)idx + 5]};
            const f3 v3 = {vertex[9 * (size_t)idx + 6], vertex[9 * (size_t)idx + 7], vertex[9 * (size_t)idx + 8]};
            const f3 center = f3_div(f3_add(f3_add(v1, v2), v3), 3.0f);
            const f3 cp = {campos[0], campos[1], campos[2]};
            f3 dsh;
            rgb_from_sh_backward(idx, D, M, center, cp, shs, s->clamped, (const f3 *)dL_dfeature, (f3 *)dL_dshs, &dsh);
            const f3 third = f3_div(dsh, 3.0f);
            dL_dv1 = f3_add(dL_dv1, third); dL_dv2 = f3_add(dL_dv2, third); dL_dv3 = f3_add(dL_dv3, third);
        }
        float *o = dL_dvertex + 9 * (size_t)idx;
        o[0] = dL_dv1.x; o[1] = dL_dv1.y; o[2] = dL_dv1.z;
        o[3] = dL_dv2.x; o[4] = dL_dv2.y; o[5] = dL_dv2.z;
        o[6] = dL_dv3.x; o[7] = dL_dv3.y; o[8] = dL_dv3.z;
        const f3 dcv = xform_vec_4x3(f3_add(f3_add(dL_dv1, dL_dv2), dL_dv3), view);
        dL_dcenter2D[2 * (size_t)idx] = dcv.x;
        dL_dcenter2D[2 * (size_t)idx + 1] = dcv.y;
    }
}

/*
 * Forward entry.  Mirrors rasterizeTrianglesForward (R2D/src/extension_interface.cu:19-152) +
 * Rasterizer::forward (R2D/src/rasterizer.cu:101-267).  Outputs are caller-allocated and are
 * zero-filled here as the reference does (:103-116).  Returns 0 on success.
 */
int ts2d_oracle_forward(int W, int H, float tan_fovx, float tan_fovy, const float *view, const float *proj,
                        const float *campos, int P, int D, int M, int C, int use_shs, float gamma,
                        float background_depth, const float *background, const float *vertex,
                        const float *shs, const float *feature, const float *opacity, int back_culling,
                        int rich_info, float *out_feature, int *radii, float *out_depth, float *out_normal,
                        float *contrib_sum, float *contrib_max, int variant, ts2d_oracle_state **state_out)
{
    if (C > TS_MAX_CHANNELS || C < 0) return 1;
    if (gamma < 0.0f) return 2;
    ts2d_oracle_state *s = (ts2d_oracle_state *)zalloc(sizeof(*s));
    s->W = W; s->H = H; s->P = P; s->C = C; s->rich_info = rich_info;
    s->variant = (variant == 3) ? 3 : 2; s->tan_fovx = tan_fovx; s->tan_fovy = tan_fovy;
    s->grid_x = (W + TS_BLOCK_X - 1) / TS_BLOCK_X;
    s->grid_y = (H + TS_BLOCK_Y - 1) / TS_BLOCK_Y;
    const size_t npix = (size_t)W * H, nt = (size_t)s->grid_x * s->grid_y;
    memset(out_feature, 0, sizeof(float) * C * npix);
    memset(radii, 0, sizeof(int) * (size_t)P);
    if (rich_info)
    {
        memset(out_depth, 0, sizeof(float) * npix);
        memset(out_normal, 0, sizeof(float) * 3 * npix);
        memset(contrib_sum, 0, sizeof(float) * (size_t)P);
        memset(contrib_max, 0, sizeof(float) * (size_t)P);
    }
    s->v1_2D = zalloc(sizeof(f2) * P); s->v2_2D = zalloc(sizeof(f2) * P); s->v3_2D = zalloc(sizeof(f2) * P);
    s->area2 = zalloc(sizeof(float) * P);
    s->normal_view = zalloc(sizeof(f3) * P); s->v_depth = zalloc(sizeof(f3) * P);
    s->depth = zalloc(sizeof(float) * P); s->rgb = zalloc(sizeof(float) * 3 * P);
    s->clamped = zalloc(3 * (size_t)P);
    s->point_offsets = zalloc(sizeof(uint32_t) * P); s->tiles_touched = zalloc(sizeof(uint32_t) * P);
    s->rect_min = zalloc(sizeof(uint32_t) * 2 * P); s->rect_max = zalloc(sizeof(uint32_t) * 2 * P);
    s->ranges = zalloc(sizeof(uint32_t) * 2 * nt);
    s->n_contrib = zalloc(sizeof(uint32_t) * npix);
    s->final_T = zalloc(sizeof(float) * npix);
    if (s->variant == 3) { s->v1_view = zalloc(sizeof(f3) * P); s->v2_view = zalloc(sizeof(f3) * P); s->v3_view = zalloc(sizeof(f3) * P); }
    *state_out = s;
    if (P == 0) return 0; /* extension_interface.cu:130 */

    if (s->variant == 3) preprocess_forward_3d(s, D, M, use_shs, back_culling, view, proj, campos, vertex, shs, radii);
    else preprocess_forward(s, D, M, use_shs, back_culling, tan_fovx, tan_fovy, view, proj, campos, vertex, shs, radii);
    bin_and_sort(s);

    const float *feat = use_shs ? s->rgb : feature; /* rasterizer.cu:244 */
    double *csum = rich_info ? (double *)zalloc(sizeof(double) * P) : NULL;
    if (s->variant == 3)
        render_forward_3d(s, gamma, feat, opacity, background_depth, background, out_feature, out_depth, out_normal, csum, contrib_max);
    else
        render_forward(s, gamma, feat, opacity, background_depth, background, out_feature, out_depth, out_normal, csum, contrib_max);
    if (rich_info)
    {
        for (int i = 0; i < P; i++) contrib_sum[i] = (float)csum[i];
        free(csum);
    }
    return 0;
}

/* ---- backward ------------------------------------------------------------------------------ */

/* R2D/src/backward.cu:9-119 */
static void rgb_from_sh_backward(int idx, int deg, int max_coeffs, f3 pos, f3 campos, const float *shs,
                                 const uint8_t *clamped, const f3 *dL_dfeature, f3 *dL_dshs, f3 *dL_dpos)
{
    f3 dir_orig = f3_sub(pos, campos);
    f3 dir = f3_div(dir_orig, f3_norm(dir_orig));
    const f3 *sh = ((const f3 *)shs) + (size_t)idx * max_coeffs;
    f3 dL_dRGB = dL_dfeature[idx];
    dL_dRGB.x *= clamped[3 * idx + 0] ? 0 : 1;
    dL_dRGB.y *= clamped[3 * idx + 1] ? 0 : 1;
    dL_dRGB.z *= clamped[3 * idx + 2] ? 0 : 1;
    f3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
    float x = dir.x, y = dir.y, z = dir.z;
    f3 *dL_dsh = dL_dshs + (size_t)idx * max_coeffs;
    dL_dsh[0] = f3_lscale(SH_C0, dL_dRGB);
    if (deg > 0)
    {
        dL_dsh[1] = f3_lscale(-SH_C1 * y, dL_dRGB);
        dL_dsh[2] = f3_lscale(SH_C1 * z, dL_dRGB);
        dL_dsh[3] = f3_lscale(-SH_C1 * x, dL_dRGB);
        dRGBdx = f3_lscale(-SH_C1, sh[3]);
        dRGBdy = f3_lscale(-SH_C1, sh[1]);
        dRGBdz = f3_lscale(SH_C1, sh[2]);
        if (deg > 1)
        {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            dL_dsh[4] = f3_lscale(SH_C2[0] * xy, dL_dRGB);
            dL_dsh[5] = f3_lscale(SH_C2[1] * yz, dL_dRGB);
            dL_dsh[6] = f3_lscale(SH_C2[2] * (2.f * zz - xx - yy), dL_dRGB);
            dL_dsh[7] = f3_lscale(SH_C2[3] * xz, dL_dRGB);
            dL_dsh[8] = f3_lscale(SH_C2[4] * (xx - yy), dL_dRGB);
            /* backward.cu:66-68; float3 sums evaluated left to right */
            f3 t;
            t = f3_lscale(SH_C2[0] * y, sh[4]);
            t = f3_add(t, f3_lscale(SH_C2[2] * 2.f * -x, sh[6]));
            t = f3_add(t, f3_lscale(SH_C2[3] * z, sh[7]));
            t = f3_add(t, f3_lscale(SH_C2[4] * 2.f * x, sh[8]));
            dRGBdx = f3_add(dRGBdx, t);
            t = f3_lscale(SH_C2[0] * x, sh[4]);
            t = f3_add(t, f3_lscale(SH_C2[1] * z, sh[5]));
            t = f3_add(t, f3_lscale(SH_C2[2] * 2.f * -y, sh[6]));
            t = f3_add(t, f3_lscale(SH_C2[4] * 2.f * -y, sh[8]));
            dRGBdy = f3_add(dRGBdy, t);
            t = f3_lscale(SH_C2[1] * y, sh[5]);
            t = f3_add(t, f3_lscale(SH_C2[2] * 2.f * 2.f * z, sh[6]));
            t = f3_add(t, f3_lscale(SH_C2[3] * x, sh[7]));
            dRGBdz = f3_add(dRGBdz, t);
            if (deg > 2)
            {
                dL_dsh[9] = f3_lscale(SH_C3[0] * y * (3.f * xx - yy), dL_dRGB);
                dL_dsh[10] = f3_lscale(SH_C3[1] * xy * z, dL_dRGB);
                dL_dsh[11] = f3_lscale(SH_C3[2] * y * (4.f * zz - xx - yy), dL_dRGB);
                dL_dsh[12] = f3_lscale(SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), dL_dRGB);
                dL_dsh[13] = f3_lscale(SH_C3[4] * x * (4.f * zz - xx - yy), dL_dRGB);
                dL_dsh[14] = f3_lscale(SH_C3[5] * z * (xx - yy), dL_dRGB);
                dL_dsh[15] = f3_lscale(SH_C3[6] * x * (xx - 3.f * yy), dL_dRGB);
                /* backward.cu:87-107; `c * sh[k] * s1 * s2 ...` is ((c*sh)*s1)*s2, sums left to right */
                t = f3_scale(f3_scale(f3_scale(f3_lscale(SH_C3[0], sh[9]), 3.f), 2.f), xy);
                t = f3_add(t, f3_scale(f3_lscale(SH_C3[1], sh[10]), yz));
                t = f3_add(t, f3_scale(f3_scale(f3_lscale(SH_C3[2], sh[11]), -2.f), xy));
                t = f3_add(t, f3_scale(f3_scale(f3_scale(f3_lscale(SH_C3[3], sh[12]), -3.f), 2.f), xz));
                t = f3_add(t, f3_scale(f3_lscale(SH_C3[4], sh[13]), (-3.f * xx + 4.f * zz - yy)));
                t = f3_add(t, f3_scale(f3_scale(f3_lscale(SH_C3[5], sh[14]), 2.f), xz));
                t = f3_add(t, f3_scale(f3_scale(f3_lscale(SH_C3[6], sh[15]), 3.f), (xx - yy)));
                dRGBdx = f3_add(dRGBdx, t);
                t = f3_scale(f3_scale(f3_lscale(SH_C3[0], sh[9]), 3.f), (xx - yy));
                t = f3_add(t, f3_scale(f3_lscale(SH_C3[1], sh[10]), xz));
                t = f3_add(t, f3_scale(f3_lscale(SH_C3[2], sh[11]), (-3.f * yy + 4.f * zz - xx)));
                t = f3_add(t, f3_scale(f3_scale(f3_scale(f3_lscale(SH_C3[3], sh[12]), -3.f), 2.f), yz));
                t = f3_add(t, f3_scale(f3_scale(f3_lscale(SH_C3[4], sh[13]), -2.f), xy));
                t = f3_add(t, f3_scale(f3_scale(f3_lscale(SH_C3[5], sh[14]), -2.f), yz));
                t = f3_add(t, f3_scale(f3_scale(f3_scale(f3_lscale(SH_C3[6], sh[15]), -3.f), 2.f), xy));
                dRGBdy = f3_add(dRGBdy, t);
                t = f3_scale(f3_lscale(SH_C3[1], sh[10]), xy);
                t = f3_add(t, f3_scale(f3_scale(f3_scale(f3_lscale(SH_C3[2], sh[11]), 4.f), 2.f), yz));
                t = f3_add(t, f3_scale(f3_scale(f3_lscale(SH_C3[3], sh[12]), 3.f), (2.f * zz - xx - yy)));
                t = f3_add(t, f3_scale(f3_scale(f3_scale(f3_lscale(SH_C3[4], sh[13]), 4.f), 2.f), xz));
                t = f3_add(t, f3_scale(f3_lscale(SH_C3[5], sh[14]), (xx - yy)));
                dRGBdz = f3_add(dRGBdz, t);
            }
        }
    }
    f3 dL_ddir = {f3_dot(dL_dRGB, dRGBdx), f3_dot(dL_dRGB, dRGBdy), f3_dot(dL_dRGB, dRGBdz)};
    *dL_dpos = dnormvdv3(dir_orig, dL_ddir);
}

/* R2D/src/backward.cu:121-129 */
static f3 project_point_backward(f3 p, const float *proj, f3 dL_dp_proj)
{
    const f4 h = xform_point_4x4(p, proj);
    const float w_inv = 1.0f / (fabsf(h.w) + TS_EPS);
    const f3 pp = {h.x * w_inv, h.y * w_inv, h.z * w_inv};
    const float aw = fabsf(w_inv);
    const f4 dh = {aw * dL_dp_proj.x, aw * dL_dp_proj.y, aw * dL_dp_proj.z, aw * (-f3_dot(dL_dp_proj, pp))};
    return xform_point_4x4_T(dh, proj);
}
/* R2D/src/backward.cu:131-142 */
static void project_vec_approx_backward(f3 p, f3 v, float tx, float ty, f2 dL_dvec_proj, f3 *dL_dp, f3 *dL_dv)
{
    const float px_pz = p.x / p.z, py_pz = p.y / p.z;
    const float vx_pz = v.x / p.z, vy_pz = v.y / p.z, vz_pz = v.z / p.z;
    const f2 d = {dL_dvec_proj.x / (p.z * tx), dL_dvec_proj.y / (p.z * ty)};
    dL_dv->x = d.x; dL_dv->y = d.y; dL_dv->z = -d.x * px_pz - d.y * py_pz;
    dL_dp->x = -d.x * vz_pz; dL_dp->y = -d.y * vz_pz;
    dL_dp->z = d.x * (2.0f * vz_pz * px_pz - vx_pz) + d.y * (2.0f * vz_pz * py_pz - vy_pz);
}

/* R2D/src/backward.cu:265-493 -- per-pixel back-to-front replay; per-triangle sums in fp64 (see header) */
static void render_backward(const ts2d_oracle_state *s, float gamma, const float *feature, const float *opacity,
                            float background_depth, const float *background, const float *dL_dout_feature,
                            const float *dL_dout_depth, const float *dL_dout_normal,
                            double *g_v2d /* P*6 */, double *g_normal /* P*3 */, double *g_vdepth /* P*3 */,
                            double *g_feature /* P*C */, double *g_opacity /* P */)
{
    const int W = s->W, H = s->H, C = s->C, rich = s->rich_info;
    const int ntiles = s->grid_x * s->grid_y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < ntiles; tile++)
    {
        const int tx = tile % s->grid_x, ty = tile / s->grid_x;
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < TS_BLOCK_Y; ly++)
            for (int lx = 0; lx < TS_BLOCK_X; lx++)
            {
                const uint32_t px = tx * TS_BLOCK_X + lx, py = ty * TS_BLOCK_Y + ly;
                if (!(px < (uint32_t)W && py < (uint32_t)H)) continue;
                const uint32_t pix_id = W * py + px;
                const f2 pixf = {(float)px, (float)py};
                float T = s->final_T[pix_id];
                const uint32_t last_contributor = s->n_contrib[pix_id];
                uint32_t contributor = r1 - r0;
                float accum_feature[TS_MAX_CHANNELS] = {0, 0, 0};
                f3 accum_normal = {0, 0, 0};
                float accum_depth = background_depth;
                float dL_dfeature_pixel[TS_MAX_CHANNELS] = {0, 0, 0};
                f3 dL_dnormal_pixel = {0, 0, 0};
                float dL_ddepth_pixel = 0;
                for (int i = 0; i < C; i++)
                {
                    accum_feature[i] = background[i];
                    dL_dfeature_pixel[i] = dL_dout_feature[(size_t)i * H * W + pix_id];
                }
                if (rich)
                {
                    dL_dnormal_pixel.x = dL_dout_normal[pix_id];
                    dL_dnormal_pixel.y = dL_dout_normal[(size_t)W * H + pix_id];
                    dL_dnormal_pixel.z = dL_dout_normal[2 * (size_t)W * H + pix_id];
                    dL_ddepth_pixel = dL_dout_depth[pix_id];
                }
                for (uint32_t kk = r1; kk > r0; kk--)
                {
                    contributor--;
                    if (contributor >= last_contributor) continue;
                    const uint32_t id = s->vals[kk - 1];
                    const float area2 = s->area2[id];
                    const f2 v1_2D = s->v1_2D[id], v2_2D = s->v2_2D[id], v3_2D = s->v3_2D[id];
                    const f2 p_v1 = f2_sub(v1_2D, pixf), p_v2 = f2_sub(v2_2D, pixf), p_v3 = f2_sub(v3_2D, pixf);
                    const float a1 = f2_cross(p_v2, p_v3) / area2;
                    const float a2 = f2_cross(p_v3, p_v1) / area2;
                    const float a3 = 1.0f - a1 - a2;
                    const float ecc = 1.0f - 3.0f * fminf(fminf(a1, a2), a3);
                    if (ecc < 0.0f || ecc > 10.0f) continue;
                    const float power = -0.5f * powf(ecc, 2.0f * gamma);
                    const float op = opacity[id];
                    const float G = expf(power);
                    const float alpha = fminf(0.99f, op * G);
                    if (alpha < 1.0f / 255.0f) continue;

                    T /= (1.0f - alpha);
                    const float contrib = alpha * T;

                    float dL_dcontrib = 0.0f;
                    f3 dL_da = {0, 0, 0};
                    for (int ch = 0; ch < C; ch++)
                    {
                        atomic_add_d(&g_feature[(size_t)id * C + ch], (double)(dL_dfeature_pixel[ch] * contrib));
                        const float feat = feature[id * C + ch];
                        dL_dcontrib += dL_dfeature_pixel[ch] * (feat - accum_feature[ch]);
                        accum_feature[ch] = alpha * feat + (1.0f - alpha) * accum_feature[ch];
                    }
                    if (rich)
                    {
                        atomic_add_d(&g_normal[3 * (size_t)id + 0], (double)(dL_dnormal_pixel.x * contrib));
                        atomic_add_d(&g_normal[3 * (size_t)id + 1], (double)(dL_dnormal_pixel.y * contrib));
                        atomic_add_d(&g_normal[3 * (size_t)id + 2], (double)(dL_dnormal_pixel.z * contrib));
                        const f3 normal = s->normal_view[id];
                        dL_dcontrib += f3_dot(dL_dnormal_pixel, f3_sub(normal, accum_normal));
                        accum_normal = f3_add(f3_lscale(alpha, normal), f3_lscale(1.0f - alpha, accum_normal));

                        const float dL_ddepth = dL_ddepth_pixel * contrib;
                        atomic_add_d(&g_vdepth[3 * (size_t)id + 0], (double)(dL_ddepth * a1));
                        atomic_add_d(&g_vdepth[3 * (size_t)id + 1], (double)(dL_ddepth * a2));
                        atomic_add_d(&g_vdepth[3 * (size_t)id + 2], (double)(dL_ddepth * a3));
                        const f3 v_depth = s->v_depth[id];
                        dL_da = f3_add(dL_da, f3_lscale(dL_ddepth, v_depth));
                        const float depth = v_depth.x * a1 + v_depth.y * a2 + v_depth.z * a3;
                        dL_dcontrib += dL_ddepth_pixel * (depth - accum_depth);
                        accum_depth = alpha * depth + (1.0f - alpha) * accum_depth;
                    }
                    const float dL_dalpha = dL_dcontrib * T;
                    float dL_dpower = 0.0f;
                    if (op * G < 0.99f) dL_dpower = dL_dalpha * alpha;       /* backward.cu:443-446 */
                    const float dL_decc = dL_dpower * 2 * gamma * power / (ecc + TS_EPS); /* :447 */

                    f3 decc_da = {0, 0, 0};                                   /* :449-461 */
                    if (a1 <= a2 && a1 <= a3) decc_da.x = -3.0f;
                    else if (a2 <= a1 && a2 <= a3) decc_da.y = -3.0f;
                    else decc_da.z = -3.0f;
                    dL_da = f3_add(dL_da, f3_lscale(dL_decc, decc_da));

                    const f2 v1_v2 = f2_sub(v2_2D, v1_2D), v2_v3 = f2_sub(v3_2D, v2_2D), v3_v1 = f2_sub(v1_2D, v3_2D);
                    const float area2_inv = 1.0f / area2;
                    const f2 da1_dv1 = f2_scale(f2_perp(f2_scale(v2_v3, a1)), area2_inv);
                    const f2 da1_dv2 = f2_scale(f2_perp(f2_add(f2_scale(v3_v1, a1), p_v3)), area2_inv);
                    const f2 da1_dv3 = f2_scale(f2_perp(f2_sub(f2_scale(v1_v2, a1), p_v2)), area2_inv);
                    const f2 da2_dv1 = f2_scale(f2_perp(f2_sub(f2_scale(v2_v3, a2), p_v3)), area2_inv);
                    const f2 da2_dv2 = f2_scale(f2_perp(f2_scale(v3_v1, a2)), area2_inv);
                    const f2 da2_dv3 = f2_scale(f2_perp(f2_add(f2_scale(v1_v2, a2), p_v1)), area2_inv);
                    const f2 da3_dv1 = f2_scale(f2_perp(f2_add(f2_scale(v2_v3, a3), p_v2)), area2_inv);
                    const f2 da3_dv2 = f2_scale(f2_perp(f2_sub(f2_scale(v3_v1, a3), p_v1)), area2_inv);
                    const f2 da3_dv3 = f2_scale(f2_perp(f2_scale(v1_v2, a3)), area2_inv);
                    const f2 g1 = f2_add(f2_add(f2_lscale(dL_da.x, da1_dv1), f2_lscale(dL_da.y, da2_dv1)), f2_lscale(dL_da.z, da3_dv1));
                    const f2 g2 = f2_add(f2_add(f2_lscale(dL_da.x, da1_dv2), f2_lscale(dL_da.y, da2_dv2)), f2_lscale(dL_da.z, da3_dv2));
                    const f2 g3 = f2_add(f2_add(f2_lscale(dL_da.x, da1_dv3), f2_lscale(dL_da.y, da2_dv3)), f2_lscale(dL_da.z, da3_dv3));
                    atomic_add_d(&g_v2d[6 * (size_t)id + 0], (double)g1.x);
                    atomic_add_d(&g_v2d[6 * (size_t)id + 1], (double)g1.y);
                    atomic_add_d(&g_v2d[6 * (size_t)id + 2], (double)g2.x);
                    atomic_add_d(&g_v2d[6 * (size_t)id + 3], (double)g2.y);
                    atomic_add_d(&g_v2d[6 * (size_t)id + 4], (double)g3.x);
                    atomic_add_d(&g_v2d[6 * (size_t)id + 5], (double)g3.y);
                    atomic_add_d(&g_opacity[id], (double)(dL_dalpha * G));   /* :490, not gated by the clamp */
                }
            }
    }
}

/* R2D/src/backward.cu:144-263 */
static void preprocess_backward(const ts2d_oracle_state *s, int D, int M, int use_shs, float tan_fovx, float tan_fovy,
                                const float *view, const float *proj, const float *campos, const float *vertex,
                                const float *shs, const int *radii, const float *g_v2d, const float *g_normal,
                                const float *g_vdepth, const float *dL_dfeature, float *dL_dvertex,
                                float *dL_dcenter2D, float *dL_dshs)
{
    const int W = s->W, H = s->H, P = s->P, rich = s->rich_info;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++)
    {
        if (radii[idx] <= 0) continue;
        const f3 v1 = {vertex[9 * (size_t)idx], vertex[9 * (size_t)idx + 1], vertex[9 * (size_t)idx + 2]};
        const f3 v2 = {vertex[9 * (size_t)idx + 3], vertex[9 * (size_t)idx + 4], vertex[9 * (size_t)idx + 5]};
        const f3 v3 = {vertex[9 * (size_t)idx + 6], vertex[9 * (size_t)idx + 7], vertex[9 * (size_t)idx + 8]};
        const f3 center = f3_div(f3_add(f3_add(v1, v2), v3), 3.0f);
        const f3 center_view = xform_point_4x3(center, view);
        const float limx = 1.3f * tan_fovx * center_view.z;
        const float limy = 1.3f * tan_fovy * center_view.z;
        const f3 cvc = {fminf(fmaxf(-limx, center_view.x), limx), fminf(fmaxf(-limy, center_view.y), limy), center_view.z};
        const f3 r1 = f3_sub(v1, center), r2 = f3_sub(v2, center), r3 = f3_sub(v3, center);
        const f3 r1_view = xform_vec_4x3(r1, view), r2_view = xform_vec_4x3(r2, view), r3_view = xform_vec_4x3(r3, view);
        const f2 r1_proj = project_vec_approx(cvc, r1_view, tan_fovx, tan_fovy);
        const f2 r2_proj = project_vec_approx(cvc, r2_view, tan_fovx, tan_fovy);
        const f2 r3_proj = project_vec_approx(cvc, r3_view, tan_fovx, tan_fovy);

        const f2 g1 = {g_v2d[6 * (size_t)idx + 0], g_v2d[6 * (size_t)idx + 1]};
        const f2 g2 = {g_v2d[6 * (size_t)idx + 2], g_v2d[6 * (size_t)idx + 3]};
        const f2 g3 = {g_v2d[6 * (size_t)idx + 4], g_v2d[6 * (size_t)idx + 5]};
        const f2 dL_dcenter_2D = f2_add(f2_add(g1, g2), g3);

        const f2 scaling = {0.5f * W, 0.5f * H};
        const float kernel_size = 0.5f;
        const f2 dL_dr1_proj = f2_add(f2_mul(scaling, g1), f2_lscale(kernel_size, dnormvdv2(r1_proj, g1)));
        const f2 dL_dr2_proj = f2_add(f2_mul(scaling, g2), f2_lscale(kernel_size, dnormvdv2(r2_proj, g2)));
        const f2 dL_dr3_proj = f2_add(f2_mul(scaling, g3), f2_lscale(kernel_size, dnormvdv2(r3_proj, g3)));
        const f2 dL_dcenter_proj = f2_mul(scaling, dL_dcenter_2D);

        f3 dL_dr1_view, dL_dr2_view, dL_dr3_view, dc;
        f3 dL_dcenter_view = {0, 0, 0};
        project_vec_approx_backward(cvc, r1_view, tan_fovx, tan_fovy, dL_dr1_proj, &dc, &dL_dr1_view);
        dL_dcenter_view = f3_add(dL_dcenter_view, dc);
        project_vec_approx_backward(cvc, r2_view, tan_fovx, tan_fovy, dL_dr2_proj, &dc, &dL_dr2_view);
        dL_dcenter_view = f3_add(dL_dcenter_view, dc);
        project_vec_approx_backward(cvc, r3_view, tan_fovx, tan_fovy, dL_dr3_proj, &dc, &dL_dr3_view);
        dL_dcenter_view = f3_add(dL_dcenter_view, dc);
        if (center_view.x < -limx || center_view.x > limx) dL_dcenter_view.x = 0; /* :209-216 */
        if (center_view.y < -limy || center_view.y > limy) dL_dcenter_view.y = 0;

        if (rich)
        {
            const f3 dL_dnormal_view = {g_normal[3 * (size_t)idx], g_normal[3 * (size_t)idx + 1], g_normal[3 * (size_t)idx + 2]};
            const f3 dL_dv_depth = {g_vdepth[3 * (size_t)idx], g_vdepth[3 * (size_t)idx + 1], g_vdepth[3 * (size_t)idx + 2]};
            const f3 c12 = f3_cross(r1_view, r2_view);
            const f3 dL_dc12 = dnormvdv3(c12, dL_dnormal_view);
            { /* backward.cu:224-227: `a += cross(..) + make_float3(0,0,d)` */
                f3 add = f3_cross(r2_view, dL_dc12);
                f3 zz = {0, 0, dL_dv_depth.x};
                add = f3_add(add, zz);
                dL_dr1_view = f3_add(dL_dr1_view, add);
            }
            {
                f3 add = f3_cross(dL_dc12, r1_view);
                f3 zz = {0, 0, dL_dv_depth.y};
                add = f3_add(add, zz);
                dL_dr2_view = f3_add(dL_dr2_view, add);
            }
            {
                f3 zz = {0, 0, dL_dv_depth.z};
                dL_dr3_view = f3_add(dL_dr3_view, zz);
            }
            {
                f3 zz = {0, 0, dL_dv_depth.x + dL_dv_depth.y + dL_dv_depth.z};
                dL_dcenter_view = f3_add(dL_dcenter_view, zz);
            }
        }

        const f3 dcp = {dL_dcenter_proj.x, dL_dcenter_proj.y, 0};
        f3 dL_dcenter = project_point_backward(center, proj, dcp);
        dL_dcenter = f3_add(dL_dcenter, xform_vec_4x3_T(dL_dcenter_view, view));

        const f3 dL_dr1 = xform_vec_4x3_T(dL_dr1_view, view);
        const f3 dL_dr2 = xform_vec_4x3_T(dL_dr2_view, view);
        const f3 dL_dr3 = xform_vec_4x3_T(dL_dr3_view, view);

        if (use_shs)
        {
            f3 dsh;
            const f3 cp = {campos[0], campos[1], campos[2]};
            rgb_from_sh_backward(idx, D, M, center, cp, shs, s->clamped, (const f3 *)dL_dfeature, (f3 *)dL_dshs, &dsh);
            dL_dcenter = f3_add(dL_dcenter, dsh);
        }
        /* backward.cu:247-249: (2*a - b - c + d) / 3 */
        const f3 dL_dv1 = f3_div(f3_add(f3_sub(f3_sub(f3_lscale(2, dL_dr1), dL_dr2), dL_dr3), dL_dcenter), 3.0f);
        const f3 dL_dv2 = f3_div(f3_add(f3_sub(f3_sub(f3_lscale(2, dL_dr2), dL_dr1), dL_dr3), dL_dcenter), 3.0f);
        const f3 dL_dv3 = f3_div(f3_add(f3_sub(f3_sub(f3_lscale(2, dL_dr3), dL_dr1), dL_dr2), dL_dcenter), 3.0f);
        float *o = dL_dvertex + 9 * (size_t)idx;
        o[0] = dL_dv1.x; o[1] = dL_dv1.y; o[2] = dL_dv1.z;
        o[3] = dL_dv2.x; o[4] = dL_dv2.y; o[5] = dL_dv2.z;
        o[6] = dL_dv3.x; o[7] = dL_dv3.y; o[8] = dL_dv3.z;
        dL_dcenter2D[2 * (size_t)idx] = dL_dcenter_2D.x;
        dL_dcenter2D[2 * (size_t)idx + 1] = dL_dcenter_2D.y;
    }
}

/*
 * Backward entry.  Mirrors rasterizeTrianglesBackward (R2D/src/extension_interface.cu:154-260) +
 * Rasterizer::backward (R2D/src/rasterizer.cu:269-358).  Outputs caller-allocated, zero-filled here.
 * dL_dfeature is (P,C): in SH mode it receives dL/d(rgb) like the reference's (rasterizer.cu:333,353).
 */
int ts2d_oracle_backward(const ts2d_oracle_state *s, float tan_fovx, float tan_fovy, const float *view,
                         const float *proj, const float *campos, int D, int M, int use_shs, float gamma,
                         float background_depth, const float *background, const float *vertex, const float *shs,
                         const float *feature, const float *opacity, const int *radii,
                         const float *dL_dout_feature, const float *dL_dout_depth, const float *dL_dout_normal,
                         float *dL_dvertex, float *dL_dcenter2D, float *dL_dshs, float *dL_dfeature, float *dL_dopacity)
{
    const int P = s->P, C = s->C;
    memset(dL_dvertex, 0, sizeof(float) * 9 * (size_t)P);
    memset(dL_dcenter2D, 0, sizeof(float) * 2 * (size_t)P);
    memset(dL_dshs, 0, sizeof(float) * 3 * (size_t)M * P);
    memset(dL_dfeature, 0, sizeof(float) * (size_t)C * P);
    memset(dL_dopacity, 0, sizeof(float) * (size_t)P);
    if (P == 0) return 0;
    if (s->variant == 3)
    {
        double *gv3 = zalloc(sizeof(double) * 9 * P), *gn3 = zalloc(sizeof(double) * 3 * P);
        double *gf3 = zalloc(sizeof(double) * (size_t)C * P), *go3 = zalloc(sizeof(double) * P);
        const float *feat3 = use_shs ? s->rgb : feature;
        render_backward_3d(s, gamma, feat3, opacity, background_depth, background, dL_dout_feature, dL_dout_depth,
                           dL_dout_normal, gv3, gn3, gf3, go3);
        float *gvf3 = zalloc(sizeof(float) * 9 * P), *gnf3 = zalloc(sizeof(float) * 3 * P);
        for (size_t i = 0; i < 9 * (size_t)P; i++) gvf3[i] = (float)gv3[i];
        for (size_t i = 0; i < 3 * (size_t)P; i++) gnf3[i] = (float)gn3[i];
        for (size_t i = 0; i < (size_t)C * P; i++) dL_dfeature[i] = (float)gf3[i];
        for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (float)go3[i];
        preprocess_backward_3d(s, D, M, use_shs, view, campos, vertex, shs, radii, gvf3, gnf3, dL_dfeature, dL_dvertex,
                               dL_dcenter2D, dL_dshs);
        free(gv3); free(gn3); free(gf3); free(go3); free(gvf3); free(gnf3);
        return 0;
    }

    double *gv = zalloc(sizeof(double) * 6 * P), *gn = zalloc(sizeof(double) * 3 * P), *gd = zalloc(sizeof(double) * 3 * P);
    double *gf = zalloc(sizeof(double) * (size_t)C * P), *go = zalloc(sizeof(double) * P);
    const float *feat = use_shs ? s->rgb : feature; /* rasterizer.cu:308 */
    render_backward(s, gamma, feat, opacity, background_depth, background, dL_dout_feature, dL_dout_depth,
                    dL_dout_normal, gv, gn, gd, gf, go);
    float *gvf = zalloc(sizeof(float) * 6 * P), *gnf = zalloc(sizeof(float) * 3 * P), *gdf = zalloc(sizeof(float) * 3 * P);
    for (size_t i = 0; i < 6 * (size_t)P; i++) gvf[i] = (float)gv[i];
    for (size_t i = 0; i < 3 * (size_t)P; i++) { gnf[i] = (float)gn[i]; gdf[i] = (float)gd[i]; }
    for (size_t i = 0; i < (size_t)C * P; i++) dL_dfeature[i] = (float)gf[i];
    for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (float)go[i];
    preprocess_backward(s, D, M, use_shs, tan_fovx, tan_fovy, view, proj, campos, vertex, shs, radii, gvf, gnf, gdf,
                        dL_dfeature, dL_dvertex, dL_dcenter2D, dL_dshs);
    free(gv); free(gn); free(gd); free(gf); free(go); free(gvf); free(gnf); free(gdf);
    return 0;
}

/* Test hook: the SH colour polynomial alone (rgb_from_sh above, R2D/src/forward.cu:9-59) for n points, so that it
 * can be pinned against the reference's Python eval_sh (tests/golden/sh_eval.npz).  rgb_out is the clamped
 * colour (max(., 0) after +0.5), clamped_out the 3 flags per point. */
void ts2d_oracle_sh_color(int n, int deg, int M, const float *shs, const float *pos, const float *campos,
                          float *rgb_out, uint8_t *clamped_out)
{
    const f3 cp = {campos[0], campos[1], campos[2]};
    for (int i = 0; i < n; i++)
    {
        const f3 p = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        f3 c = rgb_from_sh(i, deg, M, p, cp, shs, clamped_out);
        rgb_out[3 * i] = c.x; rgb_out[3 * i + 1] = c.y; rgb_out[3 * i + 2] = c.z;
    }
}

/* ---- state introspection for tests --------------------------------------------------------- */
int64_t ts2d_oracle_num_rendered(const ts2d_oracle_state *s) { return s->N; }
int ts2d_oracle_grid_x(const ts2d_oracle_state *s) { return s->grid_x; }
int ts2d_oracle_grid_y(const ts2d_oracle_state *s) { return s->grid_y; }
/* field: 0 v1_2D(P*2 f32) 1 v2_2D 2 v3_2D 3 area2(P) 4 normal_view(P*3) 5 v_depth(P*3) 6 depth(P) 7 rgb(P*3)
 *        8 clamped(P*3 u8) 9 point_offsets(P u32) 10 tiles_touched(P u32) 11 rect_min(P*2 u32) 12 rect_max(P*2 u32)
 *        13 keys_unsorted(N u64) 14 keys(N u64) 15 vals_unsorted(N u32) 16 vals(N u32)
 *        17 ranges(T*2 u32) 18 n_contrib(W*H u32) 19 final_T(W*H f32) */
const void *ts2d_oracle_field(const ts2d_oracle_state *s, int field)
{
    switch (field)
    {
    case 0: return s->v1_2D; case 1: return s->v2_2D; case 2: return s->v3_2D; case 3: return s->area2;
    case 4: return s->normal_view; case 5: return s->v_depth; case 6: return s->depth; case 7: return s->rgb;
    case 8: return s->clamped; case 9: return s->point_offsets; case 10: return s->tiles_touched;
    case 11: return s->rect_min; case 12: return s->rect_max; case 13: return s->keys_unsorted;
    case 14: return s->keys; case 15: return s->vals_unsorted; case 16: return s->vals;
    case 17: return s->ranges; case 18: return s->n_contrib; case 19: return s->final_T;
    case 20: return s->v1_view; case 21: return s->v2_view; case 22: return s->v3_view;
    default: return NULL;
    }
}
int ts2d_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void ts2d_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
