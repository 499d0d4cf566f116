// preprocess3d.hip -- per-triangle forward / backward kernels of the 3D variant (rasterizer_type="3D").
//
// Behaviour follows FORWARD::preprocessCUDA and BACKWARD::preprocessCUDA of the reference's
// submodules/diff-triangle-rasterization-3D ("R3D": src/forward.cu:60-146, src/backward.cu:144-214).  Differences to
// the 2D variant: vertices are transformed to view space individually; a triangle is culled when
// ANY vertex of its 3x-dilated copy is behind the near plane; the tile rectangle comes from the projected dilated
// vertices (projToPix, fp32); the normal is NOT normalised; dL_dcenter2D is the view-space xy of the summed vertex
// gradient.  Built with -ffp-contract=off like preprocess.hip (bit-comparable integer state).
//
// Render record of the 3D variant (16 floats = 64 B): v1_view.xyz v2_view.xyz v3_view.xyz normal_view.xyz opacity r g b
// Gradient record (16 floats): dL/dv1_view dL/dv2_view dL/dv3_view dL/dnormal_view dL/dopacity dL/drgb
#include "ts2d_common.h"
#include "ts2d_math.h"
#include "ts2d_sh.h"
#include "ts2d_preprocess_launch.h"

using namespace ts;

namespace
{
__device__ __forceinline__ float proj_to_pix(float v, int S) { return (v + 1.0f) * S * 0.5f - 0.5f; } // R3D auxiliary.h:35-38

// One triangle; `vp` / `shp` = its vertex / SH rows (global memory or LDS, ts2d_preprocess_launch.h).
template <int MODE> // PRE_ALL / PRE_NOCOLOUR, see preprocess.hip
__device__ __forceinline__ void preprocess3d_fwd_one(const PreprocessArgs &a, int32_t *__restrict__ radii, const GeometryStateView &g,
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
    do
    {
        const f3 v1_view = xform_point_4x3(v1, a.viewmatrix), v2_view = xform_point_4x3(v2, a.viewmatrix),
                 v3_view = xform_point_4x3(v3, a.viewmatrix);
        const f3 center_view = divf(add(add(v1_view, v2_view), v3_view), 3.0f);
        const f3 normal_view = cross(sub(v2_view, v1_view), sub(v3_view, v1_view));
        if (norm(normal_view) < TS_EPS) break;                 // R3D forward.cu:99
        if (a.back_culling && normal_view.z >= 0) break;       // R3D forward.cu:101

        const float dilation = 3.0f;
        const f3 center = divf(add(add(v1, v2), v3), 3.0f);
        const f3 d1 = add(center, scale(dilation, sub(v1, center)));
        const f3 d2 = add(center, scale(dilation, sub(v2, center)));
        const f3 d3 = add(center, scale(dilation, sub(v3, center)));
        const f3 p1 = project_point(d1, a.projmatrix), p2 = project_point(d2, a.projmatrix), p3 = project_point(d3, a.projmatrix);
        if (p1.z <= 0 || p2.z <= 0 || p3.z <= 0) break;        // near culling, R3D forward.cu:114

        const f2 q1 = {proj_to_pix(p1.x, a.W), proj_to_pix(p1.y, a.H)};
        const f2 q2 = {proj_to_pix(p2.x, a.W), proj_to_pix(p2.y, a.H)};
        const f2 q3 = {proj_to_pix(p3.x, a.W), proj_to_pix(p3.y, a.H)};
        const f2 v_min = {fminf(fminf(q1.x, q2.x), q3.x), fminf(fminf(q1.y, q2.y), q3.y)};
        const f2 v_max = {fmaxf(fmaxf(q1.x, q2.x), q3.x), fmaxf(fmaxf(q1.y, q2.y), q3.y)};
        const int rminx = min(a.grid_x, max(0, f2i(v_min.x / TS_TILE)));
        const int rminy = min(a.grid_y, max(0, f2i(v_min.y / TS_TILE)));
        const int rmaxx = min(a.grid_x, max(0, f2i((v_max.x + TS_TILE - 1) / TS_TILE)));
        const int rmaxy = min(a.grid_y, max(0, f2i((v_max.y + TS_TILE - 1) / TS_TILE)));
        if (rmaxx <= rminx || rmaxy <= rminy) break;

        f3 rgb = {0, 0, 0};
        if (MODE == PRE_NOCOLOUR) {}
        else if (a.use_shs)
        {
            const f3 cp = {a.campos[0], a.campos[1], a.campos[2]};
            rgb = sh_to_rgb(a.D, shp, center, cp);
            out_clamped = (uint8_t)((rgb.x < 0 ? 1 : 0) | (rgb.y < 0 ? 2 : 0) | (rgb.z < 0 ? 4 : 0));
            rgb = {fmaxf(rgb.x, 0.0f), fmaxf(rgb.y, 0.0f), fmaxf(rgb.z, 0.0f)};
        }
        else
        {
            const float *fp = a.feature + (size_t)idx * a.C;
            rgb.x = a.C > 0 ? fp[0] : 0.0f;
            rgb.y = a.C > 1 ? fp[1] : 0.0f;
            rgb.z = a.C > 2 ? fp[2] : 0.0f;
        }
        rec[0] = v1_view.x; rec[1] = v1_view.y; rec[2] = v1_view.z;
        rec[3] = v2_view.x; rec[4] = v2_view.y; rec[5] = v2_view.z;
        rec[6] = v3_view.x; rec[7] = v3_view.y; rec[8] = v3_view.z;
        rec[9] = normal_view.x; rec[10] = normal_view.y; rec[11] = normal_view.z;
        rec[12] = a.opacity[idx];
        rec[13] = rgb.x; rec[14] = rgb.y; rec[15] = rgb.z;
        out_depth = center_view.z;
        out_tiles = (uint32_t)(rmaxx - rminx) * (uint32_t)(rmaxy - rminy);
        out_rect = {(uint32_t)rminx | ((uint32_t)rminy << 16), (uint32_t)rmaxx | ((uint32_t)rmaxy << 16)};
        out_radius = f2i(fmaxf(ceilf((v_max.x - v_min.x) * 0.5f), ceilf((v_max.y - v_min.y) * 0.5f)));
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

// One triangle; `ov` (9 floats) / `osh` (3 M floats, may be null) receive dL_dvertex / dL_dshs (global memory or LDS rows).
__device__ __forceinline__ f3 preprocess3d_bwd_one(const PreprocessArgs &a, const int32_t *__restrict__ radii,
                                                     const GeometryStateView &g, const float *__restrict__ grad_rec, int idx,
                                                     const float *vp, const float *shp, float *ov, float *osh,
                                                     float *__restrict__ dL_dcenter2D, float *__restrict__ dL_dfeature,
                                                     float *__restrict__ dL_dopacity)
{
    float *oc = dL_dcenter2D + 2 * (size_t)idx;
    if (radii[idx] <= 0) // R3D backward.cu:165
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
    const f3 v1 = {vp[0], vp[1], vp[2]}, v2 = {vp[3], vp[4], vp[5]}, v3 = {vp[6], vp[7], vp[8]}; // before ov (may alias vp) is written
    const float4 *rp = g.rec + 4 * (size_t)idx;
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
    const f3 v1_view = {r0.x, r0.y, r0.z}, v2_view = {r0.w, r1.x, r1.y}, v3_view = {r1.z, r1.w, r2.x};
    const float4 *gr = (const float4 *)(grad_rec + TS_GRAD_FLOATS * (size_t)idx);
    const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3];
    f3 gv1 = {g0.x, g0.y, g0.z}, gv2 = {g0.w, g1.x, g1.y}, gv3 = {g1.z, g1.w, g2.x};
    const f3 gn = {g2.y, g2.z, g2.w};
    const float gop = g3.x;
    const f3 grgb = {g3.y, g3.z, g3.w};
    f3 grgb_out = grgb;

    gv1 = add(gv1, cross(sub(v2_view, v3_view), gn)); // R3D backward.cu:176-178
    gv2 = add(gv2, cross(sub(v3_view, v1_view), gn));
    gv3 = add(gv3, cross(sub(v1_view, v2_view), gn));
    f3 dL_dv1 = xform_vec_4x3_T(gv1, a.viewmatrix), dL_dv2 = xform_vec_4x3_T(gv2, a.viewmatrix),
       dL_dv3 = xform_vec_4x3_T(gv3, a.viewmatrix);
    f3 masked = {0.0f, 0.0f, 0.0f}; // the clamp-masked colour gradient: returned, for callers that expand dL_dshs themselves
    if (a.use_shs)
    {
        const f3 center = divf(add(add(v1, v2), v3), 3.0f);
        const uint8_t cl = g.clamped[idx];
        f3 dL_dRGB = grgb;
        dL_dRGB.x *= (cl & 1) ? 0.0f : 1.0f;
        dL_dRGB.y *= (cl & 2) ? 0.0f : 1.0f;
        dL_dRGB.z *= (cl & 4) ? 0.0f : 1.0f;
        const f3 cp = {a.campos[0], a.campos[1], a.campos[2]};
        // when osh aliases shp (LDS row) the coefficients must be consumed before the gradients are written
        const f3 dsh = sh_backward(a.D, a.M, shp, center, cp, dL_dRGB, nullptr);
        masked = dL_dRGB;
        if (osh) sh_grad_store(a.D, a.M, center, cp, dL_dRGB, osh);
        if (!osh) grgb_out = dL_dRGB; // factored exchange (TS2D_FLAG_SH_FACTORED)
        const f3 third = divf(dsh, 3.0f); // R3D backward.cu:196-198
        dL_dv1 = add(dL_dv1, third); dL_dv2 = add(dL_dv2, third); dL_dv3 = add(dL_dv3, third);
    }
    ov[0] = dL_dv1.x; ov[1] = dL_dv1.y; ov[2] = dL_dv1.z;
    ov[3] = dL_dv2.x; ov[4] = dL_dv2.y; ov[5] = dL_dv2.z;
    ov[6] = dL_dv3.x; ov[7] = dL_dv3.y; ov[8] = dL_dv3.z;
    const f3 dcv = xform_vec_4x3(add(add(dL_dv1, dL_dv2), dL_dv3), a.viewmatrix); // R3D backward.cu:211-213
    oc[0] = dcv.x; oc[1] = dcv.y;
    dL_dopacity[idx] = gop;
    float *of = dL_dfeature + (size_t)idx * a.C;
    if (a.C > 0) of[0] = grgb_out.x;
    if (a.C > 1) of[1] = grgb_out.y;
    if (a.C > 2) of[2] = grgb_out.z;
    return masked;
}
struct Raster3D
{
    template <int MODE, class... T> static __device__ __forceinline__ void fwd(T... t) { preprocess3d_fwd_one<MODE>(t...); }
    template <class... T> static __device__ __forceinline__ f3 bwd(T... t) { return preprocess3d_bwd_one(t...); }
};
} // namespace

void ts_launch_preprocess3d_fwd(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s, int mode)
{
    launch_preprocess_fwd<Raster3D>(a, radii, g, s, mode);
}

void ts_launch_preprocess3d_bwd(const PreprocessArgs &a, const int32_t *radii, const GeometryStateView &g,
                                const float *grad_rec, float *dL_dvertex, float *dL_dcenter2D, float *dL_dshs,
                                float *dL_dfeature, float *dL_dopacity, hipStream_t s)
{
    launch_preprocess_bwd<Raster3D>(a, radii, g, grad_rec, dL_dvertex, dL_dcenter2D, dL_dshs, dL_dfeature, dL_dopacity, s);
}
