// ts2d_sh.h -- spherical-harmonics colour (degree 0..3) and its backward, shared by the 2D and 3D preprocess kernels.
// Same polynomial in both reference rasterizers (R2D/src/forward.cu:9-59 == R3D/src/forward.cu:9-58; backward
// R2D/src/backward.cu:9-119 == R3D/src/backward.cu:9-118).  Evaluated in the reference's expression order; the including
// translation units are built with -ffp-contract=off.
#pragma once
#include "ts2d_math.h"

namespace ts
{
__device__ __forceinline__ f3 ld3(const float *p) { return {p[0], p[1], p[2]}; }

// SH -> RGB at direction (pos - campos); forward.cu:9-59.  Returns the unclamped colour + 0.5.
__device__ __forceinline__ f3 sh_to_rgb(int deg, const float *sh, f3 pos, f3 campos)
{
    f3 dir = sub(pos, campos);
    dir = divf(dir, norm(dir));
    f3 rgb = scale(SH_C0, ld3(sh));
    if (deg > 0)
    {
        const float x = dir.x, y = dir.y, z = dir.z;
        rgb = sub(add(sub(rgb, scale(SH_C1 * y, ld3(sh + 3))), scale(SH_C1 * z, ld3(sh + 6))), scale(SH_C1 * x, ld3(sh + 9)));
        if (deg > 1)
        {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            rgb = add(rgb, scale(SH_C2_0 * xy, ld3(sh + 12)));
            rgb = add(rgb, scale(SH_C2_1 * yz, ld3(sh + 15)));
            rgb = add(rgb, scale(SH_C2_2 * (2.0f * zz - xx - yy), ld3(sh + 18)));
            rgb = add(rgb, scale(SH_C2_3 * xz, ld3(sh + 21)));
            rgb = add(rgb, scale(SH_C2_4 * (xx - yy), ld3(sh + 24)));
            if (deg > 2)
            {
                rgb = add(rgb, scale(SH_C3_0 * y * (3.0f * xx - yy), ld3(sh + 27)));
                rgb = add(rgb, scale(SH_C3_1 * xy * z, ld3(sh + 30)));
                rgb = add(rgb, scale(SH_C3_2 * y * (4.0f * zz - xx - yy), ld3(sh + 33)));
                rgb = add(rgb, scale(SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy), ld3(sh + 36)));
                rgb = add(rgb, scale(SH_C3_4 * x * (4.0f * zz - xx - yy), ld3(sh + 39)));
                rgb = add(rgb, scale(SH_C3_5 * z * (xx - yy), ld3(sh + 42)));
                rgb = add(rgb, scale(SH_C3_6 * x * (xx - 3.0f * yy), ld3(sh + 45)));
            }
        }
    }
    rgb.x += 0.5f; rgb.y += 0.5f; rgb.z += 0.5f;
    return rgb;
}

__device__ __forceinline__ void st3(float *p, f3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// d(colour)/d(sh_k) for the active coefficients, in the reference's expression order (backward.cu:20-85: the scalar
// factor each dL_dsh[k] = factor * dL_dRGB is formed first, left to right).  Returns the number of active coefficients.
__device__ __forceinline__ int sh_basis(int deg, f3 dir, float *b)
{
    const float x = dir.x, y = dir.y, z = dir.z;
    b[0] = SH_C0;
    if (deg < 1) return 1;
    b[1] = -SH_C1 * y; b[2] = SH_C1 * z; b[3] = -SH_C1 * x;
    if (deg < 2) return 4;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2_0 * xy; b[5] = SH_C2_1 * yz; b[6] = SH_C2_2 * (2.f * zz - xx - yy); b[7] = SH_C2_3 * xz;
    b[8] = SH_C2_4 * (xx - yy);
    if (deg < 3) return 9;
    b[9] = SH_C3_0 * y * (3.f * xx - yy); b[10] = SH_C3_1 * xy * z; b[11] = SH_C3_2 * y * (4.f * zz - xx - yy);
    b[12] = SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy); b[13] = SH_C3_4 * x * (4.f * zz - xx - yy);
    b[14] = SH_C3_5 * z * (xx - yy); b[15] = SH_C3_6 * x * (xx - 3.f * yy);
    return 16;
}

// dL_dsh[k] = basis_k * dL_dRGB for the active coefficients, zeros above (backward.cu:20-85).
__device__ __forceinline__ void sh_grad_store(int deg, int M, f3 pos, f3 campos, f3 dL_dRGB, float *dL_dsh)
{
    const f3 dir_orig = sub(pos, campos);
    const f3 dir = divf(dir_orig, norm(dir_orig));
    float b[16];
    const int written = sh_basis(deg, dir, b);
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k < written) st3(dL_dsh + 3 * k, scale(b[k], dL_dRGB));
    // constant indices under a predicate (instead of a loop with run-time bounds): the row may live in registers
#pragma unroll
    for (int k = 0; k < 48; k++)
        if (k >= written * 3 && k < M * 3) dL_dsh[k] = 0.0f;
}

// backward.cu:9-119.  Writes all M coefficient gradients (zeros above the active degree) unless dL_dsh is null (the
// factored multi-GPU exchange rebuilds them from dL_dRGB, see shgrad.hip).  Returns dL/d(pos).
__device__ __forceinline__ f3 sh_backward(int deg, int M, const float *sh, f3 pos, f3 campos, f3 dL_dRGB, float *dL_dsh)
{
    const f3 dir_orig = sub(pos, campos);
    const f3 dir = divf(dir_orig, norm(dir_orig));
    f3 dRGBdx = {0, 0, 0}, dRGBdy = {0, 0, 0}, dRGBdz = {0, 0, 0};
    const float x = dir.x, y = dir.y, z = dir.z;
    if (dL_dsh) sh_grad_store(deg, M, pos, campos, dL_dRGB, dL_dsh);
    if (deg > 0)
    {
        dRGBdx = scale(-SH_C1, ld3(sh + 9));
        dRGBdy = scale(-SH_C1, ld3(sh + 3));
        dRGBdz = scale(SH_C1, ld3(sh + 6));
        if (deg > 1)
        {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            const f3 s4 = ld3(sh + 12), s5 = ld3(sh + 15), s6 = ld3(sh + 18), s7 = ld3(sh + 21), s8 = ld3(sh + 24);
            f3 t; // backward.cu:66-68, sums left to right
            t = scale(SH_C2_0 * y, s4);
            t = add(t, scale(SH_C2_2 * 2.f * -x, s6));
            t = add(t, scale(SH_C2_3 * z, s7));
            t = add(t, scale(SH_C2_4 * 2.f * x, s8));
            dRGBdx = add(dRGBdx, t);
            t = scale(SH_C2_0 * x, s4);
            t = add(t, scale(SH_C2_1 * z, s5));
            t = add(t, scale(SH_C2_2 * 2.f * -y, s6));
            t = add(t, scale(SH_C2_4 * 2.f * -y, s8));
            dRGBdy = add(dRGBdy, t);
            t = scale(SH_C2_1 * y, s5);
            t = add(t, scale(SH_C2_2 * 2.f * 2.f * z, s6));
            t = add(t, scale(SH_C2_3 * x, s7));
            dRGBdz = add(dRGBdz, t);
            if (deg > 2)
            {
                const f3 s9 = ld3(sh + 27), s10 = ld3(sh + 30), s11 = ld3(sh + 33), s12 = ld3(sh + 36), s13 = ld3(sh + 39),
                         s14 = ld3(sh + 42), s15 = ld3(sh + 45);
                // backward.cu:87-107: `c * sh * s1 * s2` is ((c*sh)*s1)*s2; sums left to right
                t = rscale(rscale(rscale(scale(SH_C3_0, s9), 3.f), 2.f), xy);
                t = add(t, rscale(scale(SH_C3_1, s10), yz));
                t = add(t, rscale(rscale(scale(SH_C3_2, s11), -2.f), xy));
                t = add(t, rscale(rscale(rscale(scale(SH_C3_3, s12), -3.f), 2.f), xz));
                t = add(t, rscale(scale(SH_C3_4, s13), (-3.f * xx + 4.f * zz - yy)));
                t = add(t, rscale(rscale(scale(SH_C3_5, s14), 2.f), xz));
                t = add(t, rscale(rscale(scale(SH_C3_6, s15), 3.f), (xx - yy)));
                dRGBdx = add(dRGBdx, t);
                t = rscale(rscale(scale(SH_C3_0, s9), 3.f), (xx - yy));
                t = add(t, rscale(scale(SH_C3_1, s10), xz));
                t = add(t, rscale(scale(SH_C3_2, s11), (-3.f * yy + 4.f * zz - xx)));
                t = add(t, rscale(rscale(rscale(scale(SH_C3_3, s12), -3.f), 2.f), yz));
                t = add(t, rscale(rscale(scale(SH_C3_4, s13), -2.f), xy));
                t = add(t, rscale(rscale(scale(SH_C3_5, s14), -2.f), yz));
                t = add(t, rscale(rscale(rscale(scale(SH_C3_6, s15), -3.f), 2.f), xy));
                dRGBdy = add(dRGBdy, t);
                t = rscale(scale(SH_C3_1, s10), xy);
                t = add(t, rscale(rscale(rscale(scale(SH_C3_2, s11), 4.f), 2.f), yz));
                t = add(t, rscale(rscale(scale(SH_C3_3, s12), 3.f), (2.f * zz - xx - yy)));
                t = add(t, rscale(rscale(rscale(scale(SH_C3_4, s13), 4.f), 2.f), xz));
                t = add(t, rscale(scale(SH_C3_5, s14), (xx - yy)));
                dRGBdz = add(dRGBdz, t);
            }
        }
    }
    const f3 dL_ddir = {dot(dL_dRGB, dRGBdx), dot(dL_dRGB, dRGBdy), dot(dL_dRGB, dRGBdz)};
    return dnormvdv(dir_orig, dL_ddir); // backward.cu:118
}

} // namespace ts
