// preprocess.hip -- per-triangle forward and backward kernels (HBM-bound; one lane per triangle).
//
// Built with -ffp-contract=off: every discrete decision here (culling, tile rectangle, radii, depth key)
// feeds integer state that the parity tests compare bit-exactly against the oracle, and the kernels are
// bandwidth-bound so FMA contraction would buy nothing.
//
// Behaviour follows FORWARD::preprocessCUDA + computeRGBFromSH (R2D/src/forward.cu:61-193, 9-59) and
// BACKWARD::preprocessCUDA + computeRGBFromSHBackward + projectPointBackward + projectVecApproxBackward
// (R2D/src/backward.cu:144-263, 9-119, 121-129, 131-142).  Layout and data movement are our own: results go
// to the 64-byte render record (ts2d_common.h) instead of eight SoA arrays, every output element is written
// (so callers never pre-zero dL_dshs & co.), and gradients arrive as one 64-byte record per triangle.
#include "ts2d_common.h"
#include "ts2d_math.h"
#include "ts2d_sh.h"
#include "ts2d_preprocess_launch.h"

using namespace ts;

namespace
{
// One triangle.  `vp` = its 9 vertex floats, `shp` = its SH row (3 M floats); either global memory or an LDS row.
// MODE (ts2d_preprocess_launch.h): PRE_ALL = everything; PRE_NOCOLOUR = everything except the SH colour (SH mode only: the record's r g b and the
// clamp flags are left 0, preprocess_colour_kernel fills them in on the library's side stream beside the ordering chain).
template <int MODE>
__device__ __forceinline__ void preprocess_fwd_one(const PreprocessArgs &a, int32_t *__restrict__ radii, const GeometryStateView &g,
                                                   int idx, const float *vp, const float *shp, float4 *rec_row)
{
    int out_radius = 0;
    uint32_t out_tiles = 0;
    uint2 out_rect = {0u, 0u};
    uint8_t out_clamped = 0;
    float out_depth = 0.0f;
    float rec[TS_REC_FLOATS];
#pragma unroll
    for (int i = 0; i < TS_REC_FLOATS; i++) rec[i] = 0.0f;

    const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]};
    const f3 center = divf(add(add(v1, v2), v3), 3.0f);
    const f3 center_proj = project_point(center, a.projmatrix);

    do
    {
        if (center_proj.z <= 0) break; // near culling, forward.cu:98

        const f3 center_view = xform_point_4x3(center, a.viewmatrix);
        const float limx = 1.3f * a.tan_fovx * center_view.z;
        const float limy = 1.3f * a.tan_fovy * center_view.z;
        const f3 cvc = {fminf(fmaxf(-limx, center_view.x), limx), fminf(fmaxf(-limy, center_view.y), limy), center_view.z};

        const f3 r1 = sub(v1, center), r2 = sub(v2, center), r3 = sub(v3, center);
        const f3 r1_view = xform_vec_4x3(r1, a.viewmatrix);
        const f3 r2_view = xform_vec_4x3(r2, a.viewmatrix);
        if (norm(cross(r1_view, r2_view)) < TS_EPS) break; // forward.cu:113

        const f3 r3_view = xform_vec_4x3(r3, a.viewmatrix);
        const f2 r1_proj = project_vec_approx(cvc, r1_view, a.tan_fovx, a.tan_fovy);
        const f2 r2_proj = project_vec_approx(cvc, r2_view, a.tan_fovx, a.tan_fovy);
        const f2 r3_proj = project_vec_approx(cvc, r3_view, a.tan_fovx, a.tan_fovy);
        const float n1 = norm(r1_proj), n2 = norm(r2_proj), n3 = norm(r3_proj);
        if (n1 < TS_EPS || n2 < TS_EPS || n3 < TS_EPS) break; // forward.cu:124

        const f2 scaling = {0.5f * a.W, 0.5f * a.H};
        const float kernel_size = 0.5f; // low-pass dilation, forward.cu:128
        const f2 r1_2D = mul(r1_proj, addf(scaling, kernel_size / n1));
        const f2 r2_2D = mul(r2_proj, addf(scaling, kernel_size / n2));
        const f2 r3_2D = mul(r3_proj, addf(scaling, kernel_size / n3));
        const f2 center_2D = {ndc2pix(center_proj.x, a.W), ndc2pix(center_proj.y, a.H)};
        const f2 v1_2D = add(center_2D, r1_2D), v2_2D = add(center_2D, r2_2D), v3_2D = add(center_2D, r3_2D);
        const float area2 = cross(sub(v2_2D, v1_2D), sub(v3_2D, v1_2D));

        if (a.back_culling) { if (area2 >= -TS_EPS) break; } // forward.cu:140-144
        else { if (fabsf(area2) < TS_EPS) break; }           // forward.cu:145-149

        const float dilation = 3.0f;
        const f2 d1 = add(center_2D, scale(dilation, r1_2D));
        const f2 d2 = add(center_2D, scale(dilation, r2_2D));
        const f2 d3 = add(center_2D, scale(dilation, r3_2D));
        const f2 v_min = {fminf(fminf(d1.x, d2.x), d3.x), fminf(fminf(d1.y, d2.y), d3.y)};
        const f2 v_max = {fmaxf(fmaxf(d1.x, d2.x), d3.x), fmaxf(fmaxf(d1.y, d2.y), d3.y)};

        // forward.cu:158-163
        const int rminx = min(a.grid_x, max(0, f2i(v_min.x / TS_TILE)));
        const int rminy = min(a.grid_y, max(0, f2i(v_min.y / TS_TILE)));
        const int rmaxx = min(a.grid_x, max(0, f2i((v_max.x + TS_TILE - 1) / TS_TILE)));
        const int rmaxy = min(a.grid_y, max(0, f2i((v_max.y + TS_TILE - 1) / TS_TILE)));
        if (rmaxx <= rminx || rmaxy <= rminy) break;

        {
        f3 rgb = {0, 0, 0};
        if (MODE == PRE_NOCOLOUR) {}
        else if (a.use_shs)
        {
            const f3 cp = {a.campos[0], a.campos[1], a.campos[2]};
            rgb = sh_to_rgb(a.D, shp, center, cp);
            out_clamped = (uint8_t)((rgb.x < 0 ? 1 : 0) | (rgb.y < 0 ? 2 : 0) | (rgb.z < 0 ? 4 : 0)); // forward.cu:55-57
            rgb = {fmaxf(rgb.x, 0.0f), fmaxf(rgb.y, 0.0f), fmaxf(rgb.z, 0.0f)};
        }
        else
        {
            const float *fp = a.feature + (size_t)idx * a.C;
            rgb.x = a.C > 0 ? fp[0] : 0.0f;
            rgb.y = a.C > 1 ? fp[1] : 0.0f;
            rgb.z = a.C > 2 ? fp[2] : 0.0f;
        }
        rec[0] = v1_2D.x; rec[1] = v1_2D.y; rec[2] = v2_2D.x; rec[3] = v2_2D.y; rec[4] = v3_2D.x; rec[5] = v3_2D.y;
        rec[6] = a.opacity[idx];
        rec[7] = rgb.x; rec[8] = rgb.y; rec[9] = rgb.z;
        if (a.rich_info) // forward.cu:173-181
        {
            f3 n_view = cross(r1_view, r2_view);
            n_view = divf(n_view, norm(n_view));
            rec[10] = n_view.x; rec[11] = n_view.y; rec[12] = n_view.z;
            rec[13] = r1_view.z + center_view.z; rec[14] = r2_view.z + center_view.z; rec[15] = r3_view.z + center_view.z;
        }
        }
        out_depth = center_view.z;
        out_tiles = (uint32_t)(rmaxx - rminx) * (uint32_t)(rmaxy - rminy);
        out_rect = {(uint32_t)rminx | ((uint32_t)rminy << 16), (uint32_t)rmaxx | ((uint32_t)rmaxy << 16)};
        out_radius = f2i(fmaxf(ceilf((v_max.x - v_min.x) * 0.5f), ceilf((v_max.y - v_min.y) * 0.5f))); // forward.cu:192
    } while (false);

    {
        radii[idx] = out_radius;
        g.tiles_touched[idx] = out_tiles;
        g.rect[idx] = out_rect;
        g.depth[idx] = out_depth;
        g.clamped[idx] = out_clamped;
        float4 *r = rec_row; // the triangle's 64-byte render record: g.rec + 4 idx, or an LDS row the workgroup writes out in one block
        r[0] = make_float4(rec[0], rec[1], rec[2], rec[3]);
        r[1] = make_float4(rec[4], rec[5], rec[6], rec[7]);
        r[2] = make_float4(rec[8], rec[9], rec[10], rec[11]);
        r[3] = make_float4(rec[12], rec[13], rec[14], rec[15]);
    }
}

// backward.cu:131-142
__device__ __forceinline__ void project_vec_approx_bwd(f3 p, f3 v, float tx, float ty, f2 dL_dvec_proj, f3 &dL_dp, f3 &dL_dv)
{
    const float px_pz = p.x / p.z, py_pz = p.y / p.z;
    const float vx_pz = v.x / p.z, vy_pz = v.y / p.z, vz_pz = v.z / p.z;
    const f2 d = {dL_dvec_proj.x / (p.z * tx), dL_dvec_proj.y / (p.z * ty)};
    dL_dv = {d.x, d.y, -d.x * px_pz - d.y * py_pz};
    dL_dp = {-d.x * vz_pz, -d.y * vz_pz, d.x * (2.0f * vz_pz * px_pz - vx_pz) + d.y * (2.0f * vz_pz * py_pz - vy_pz)};
}

// One triangle.  `vp` / `shp`: its vertex and SH rows (global or LDS); `ov` (9 floats) and `osh` (3 M floats, may be null):
// where dL_dvertex / dL_dshs of this triangle go (global or LDS rows that the caller flushes).
__device__ __forceinline__ f3 preprocess_bwd_one(const PreprocessArgs &a, const int32_t *__restrict__ radii,
                                                   const GeometryStateView &g, const float *__restrict__ grad_rec, int idx,
                                                   const float *vp, const float *shp, float *ov, float *osh,
                                                   float *__restrict__ dL_dcenter2D, float *__restrict__ dL_dfeature,
                                                   float *__restrict__ dL_dopacity)
{
    float *oc = dL_dcenter2D + 2 * (size_t)idx;
    if (radii[idx] <= 0) // backward.cu:165; the reference leaves zero-initialised outputs for these
    {
#pragma unroll
        for (int i = 0; i < 9; i++) ov[i] = 0.0f;
        oc[0] = 0.0f; oc[1] = 0.0f;
        dL_dopacity[idx] = 0.0f;
        for (int c = 0; c < a.C; c++) dL_dfeature[(size_t)idx * a.C + c] = 0.0f;
        if (a.use_shs && osh)
        {
#pragma unroll
            for (int k = 0; k < 48; k++) // constant indices: the row may live in registers (M <= 16, validate())
                if (k < a.M * 3) osh[k] = 0.0f;
        }
        return {0.0f, 0.0f, 0.0f};
    }

    const float4 *gr = (const float4 *)(grad_rec + TS_GRAD_FLOATS * (size_t)idx);
    const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3];
    const f2 dL_dv1_2D = {g0.x, g0.y}, dL_dv2_2D = {g0.z, g0.w}, dL_dv3_2D = {g1.x, g1.y};
    const float dL_dop = g1.z;
    f3 dL_drgb = {g1.w, g2.x, g2.y};
    const f3 dL_dnormal_view = {g2.z, g2.w, g3.x};
    const f3 dL_dv_depth = {g3.y, g3.z, g3.w};

    const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]};
    const f3 center = divf(add(add(v1, v2), v3), 3.0f);
    const f3 center_view = xform_point_4x3(center, a.viewmatrix);
    const float limx = 1.3f * a.tan_fovx * center_view.z;
    const float limy = 1.3f * a.tan_fovy * center_view.z;
    const f3 cvc = {fminf(fmaxf(-limx, center_view.x), limx), fminf(fmaxf(-limy, center_view.y), limy), center_view.z};
    const f3 r1 = sub(v1, center), r2 = sub(v2, center), r3 = sub(v3, center);
    const f3 r1_view = xform_vec_4x3(r1, a.viewmatrix), r2_view = xform_vec_4x3(r2, a.viewmatrix),
             r3_view = xform_vec_4x3(r3, a.viewmatrix);
    const f2 r1_proj = project_vec_approx(cvc, r1_view, a.tan_fovx, a.tan_fovy);
    const f2 r2_proj = project_vec_approx(cvc, r2_view, a.tan_fovx, a.tan_fovy);
    const f2 r3_proj = project_vec_approx(cvc, r3_view, a.tan_fovx, a.tan_fovy);

    const f2 dL_dcenter_2D = add(add(dL_dv1_2D, dL_dv2_2D), dL_dv3_2D);
    const f2 scaling = {0.5f * a.W, 0.5f * a.H};
    const float kernel_size = 0.5f;
    const f2 dL_dr1_proj = add(mul(scaling, dL_dv1_2D), scale(kernel_size, dnormvdv(r1_proj, dL_dv1_2D)));
    const f2 dL_dr2_proj = add(mul(scaling, dL_dv2_2D), scale(kernel_size, dnormvdv(r2_proj, dL_dv2_2D)));
    const f2 dL_dr3_proj = add(mul(scaling, dL_dv3_2D), scale(kernel_size, dnormvdv(r3_proj, dL_dv3_2D)));
    const f2 dL_dcenter_proj = mul(scaling, dL_dcenter_2D);

    f3 dL_dr1_view, dL_dr2_view, dL_dr3_view, dc;
    f3 dL_dcenter_view = {0, 0, 0};
    project_vec_approx_bwd(cvc, r1_view, a.tan_fovx, a.tan_fovy, dL_dr1_proj, dc, dL_dr1_view);
    dL_dcenter_view = add(dL_dcenter_view, dc);
    project_vec_approx_bwd(cvc, r2_view, a.tan_fovx, a.tan_fovy, dL_dr2_proj, dc, dL_dr2_view);
    dL_dcenter_view = add(dL_dcenter_view, dc);
    project_vec_approx_bwd(cvc, r3_view, a.tan_fovx, a.tan_fovy, dL_dr3_proj, dc, dL_dr3_view);
    dL_dcenter_view = add(dL_dcenter_view, dc);
    if (center_view.x < -limx || center_view.x > limx) dL_dcenter_view.x = 0; // backward.cu:209-216
    if (center_view.y < -limy || center_view.y > limy) dL_dcenter_view.y = 0;

    if (a.rich_info) // backward.cu:218-228
    {
        const f3 c12 = cross(r1_view, r2_view);
        const f3 dL_dc12 = dnormvdv(c12, dL_dnormal_view);
        dL_dr1_view = add(dL_dr1_view, add(cross(r2_view, dL_dc12), f3{0, 0, dL_dv_depth.x}));
        dL_dr2_view = add(dL_dr2_view, add(cross(dL_dc12, r1_view), f3{0, 0, dL_dv_depth.y}));
        dL_dr3_view = add(dL_dr3_view, f3{0, 0, dL_dv_depth.z});
        dL_dcenter_view = add(dL_dcenter_view, f3{0, 0, dL_dv_depth.x + dL_dv_depth.y + dL_dv_depth.z});
    }

    // projectPointBackward, backward.cu:121-129 (only xy of the centre gradient is propagated, :231)
    f3 dL_dcenter;
    {
        const f4 h = xform_point_4x4(center, a.projmatrix);
        const float w_inv = 1.0f / (fabsf(h.w) + TS_EPS);
        const f3 pp = {h.x * w_inv, h.y * w_inv, h.z * w_inv};
        const f3 dpp = {dL_dcenter_proj.x, dL_dcenter_proj.y, 0};
        const float aw = fabsf(w_inv);
        const f4 dh = {aw * dpp.x, aw * dpp.y, aw * dpp.z, aw * (-dot(dpp, pp))};
        dL_dcenter = xform_point_4x4_T(dh, a.projmatrix);
    }
    dL_dcenter = add(dL_dcenter, xform_vec_4x3_T(dL_dcenter_view, a.viewmatrix));

    const f3 dL_dr1 = xform_vec_4x3_T(dL_dr1_view, a.viewmatrix);
    const f3 dL_dr2 = xform_vec_4x3_T(dL_dr2_view, a.viewmatrix);
    const f3 dL_dr3 = xform_vec_4x3_T(dL_dr3_view, a.viewmatrix);

    f3 masked = {0.0f, 0.0f, 0.0f}; // the clamp-masked colour gradient: returned, for callers that expand dL_dshs themselves
    if (a.use_shs)
    {
        const uint8_t cl = g.clamped[idx];
        f3 dL_dRGB = dL_drgb; // backward.cu:19-22
        dL_dRGB.x *= (cl & 1) ? 0.0f : 1.0f;
        dL_dRGB.y *= (cl & 2) ? 0.0f : 1.0f;
        dL_dRGB.z *= (cl & 4) ? 0.0f : 1.0f;
        const f3 cp = {a.campos[0], a.campos[1], a.campos[2]};
        // when osh aliases shp (LDS row, staged kernel) the coefficients must be consumed before the gradients are written
        const f3 dsh = sh_backward(a.D, a.M, shp, center, cp, dL_dRGB, nullptr);
        masked = dL_dRGB;
        if (osh) sh_grad_store(a.D, a.M, center, cp, dL_dRGB, osh);
        if (!osh) dL_drgb = dL_dRGB; // factored exchange (TS2D_FLAG_SH_FACTORED): hand out the clamp-masked colour gradient
        dL_dcenter = add(dL_dcenter, dsh);
    }

    // backward.cu:247-249
    const f3 dL_dv1 = divf(add(sub(sub(scale(2.0f, dL_dr1), dL_dr2), dL_dr3), dL_dcenter), 3.0f);
    const f3 dL_dv2 = divf(add(sub(sub(scale(2.0f, dL_dr2), dL_dr1), dL_dr3), dL_dcenter), 3.0f);
    const f3 dL_dv3 = divf(add(sub(sub(scale(2.0f, dL_dr3), dL_dr1), dL_dr2), dL_dcenter), 3.0f);
    ov[0] = dL_dv1.x; ov[1] = dL_dv1.y; ov[2] = dL_dv1.z;
    ov[3] = dL_dv2.x; ov[4] = dL_dv2.y; ov[5] = dL_dv2.z;
    ov[6] = dL_dv3.x; ov[7] = dL_dv3.y; ov[8] = dL_dv3.z;
    oc[0] = dL_dcenter_2D.x; oc[1] = dL_dcenter_2D.y;
    dL_dopacity[idx] = dL_dop;
    float *of = dL_dfeature + (size_t)idx * a.C;
    if (a.C > 0) of[0] = dL_drgb.x;
    if (a.C > 1) of[1] = dL_drgb.y;
    if (a.C > 2) of[2] = dL_drgb.z;
    return masked;
}

struct Raster2D
{
    template <int MODE, class... T> static __device__ __forceinline__ void fwd(T... t) { preprocess_fwd_one<MODE>(t...); }
    template <class... T> static __device__ __forceinline__ f3 bwd(T... t) { return preprocess_bwd_one(t...); }
};
} // namespace

void ts_launch_preprocess_fwd(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s, int mode)
{
    launch_preprocess_fwd<Raster2D>(a, radii, g, s, mode);
}

bool ts_preprocess_fwd_splittable(const PreprocessArgs &a) { return preprocess_fwd_splittable(a); }
void ts_launch_preprocess_colour(const PreprocessArgs &a, const GeometryStateView &g, int variant, int blocks, hipStream_t s)
{
    if (variant == 3) launch_preprocess_colour<13>(a, g, blocks, s); // record layouts: ts2d_common.h
    else launch_preprocess_colour<7>(a, g, blocks, s);
}

void ts_launch_preprocess_bwd(const PreprocessArgs &a, const int32_t *radii, const GeometryStateView &g,
                              const float *grad_rec, float *dL_dvertex, float *dL_dcenter2D, float *dL_dshs,
                              float *dL_dfeature, float *dL_dopacity, hipStream_t s)
{
    launch_preprocess_bwd<Raster2D>(a, radii, g, grad_rec, dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity, s);
}
