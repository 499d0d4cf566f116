// ts2d_math.h -- device-side vector helpers and the projection model shared by the preprocess kernels.
// Behaviour follows R2D/src/auxiliary.h (cited per function).  The preprocess translation unit is built with
// -ffp-contract=off so that these expressions evaluate exactly as written (bit-comparable integer state).
#pragma once
#include <hip/hip_runtime.h>

#define TS_EPS 1e-8f // R2D/src/auxiliary.h:8

namespace ts
{
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ f2 add(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f2 sub(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ f2 mul(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f2 scale(float s, f2 a) { return {s * a.x, s * a.y}; }
__device__ __forceinline__ f2 addf(f2 a, float s) { return {a.x + s, a.y + s}; }
__device__ __forceinline__ float cross(f2 a, f2 b) { return a.x * b.y - a.y * b.x; } // auxiliary.h:174
__device__ __forceinline__ float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float norm(f2 a) { return sqrtf(dot(a, a)); }

__device__ __forceinline__ f3 add(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 scale(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ f3 rscale(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 divf(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float norm(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) // auxiliary.h:179
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// float -> int, round toward zero, saturating, NaN -> 0 (v_cvt_i32_f32 semantics == CUDA cvt.rzi.s32.f32)
__device__ __forceinline__ int f2i(float v) { return __float2int_rz(v); }

// auxiliary.h:35-38 (double arithmetic)
__device__ __forceinline__ float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:40-48
__device__ __forceinline__ f3 xform_point_4x3(f3 p, const float *m)
{
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
// auxiliary.h:50-58
__device__ __forceinline__ f4 xform_point_4x4(f3 p, const float *m)
{
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14], m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15]};
}
// auxiliary.h:60-67
__device__ __forceinline__ f3 xform_point_4x4_T(f4 p, const float *m)
{
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3] * p.w, m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7] * p.w,
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11] * p.w};
}
// auxiliary.h:69-77
__device__ __forceinline__ f3 xform_vec_4x3(f3 p, const float *m)
{
    return {m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
            m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
// auxiliary.h:79-87
__device__ __forceinline__ f3 xform_vec_4x3_T(f3 p, const float *m)
{
    return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
}
// auxiliary.h:89-95
__device__ __forceinline__ f3 project_point(f3 p, const float *proj)
{
    f4 h = xform_point_4x4(p, proj);
    float w_inv = 1.0f / (fabsf(h.w) + TS_EPS);
    return {h.x * w_inv, h.y * w_inv, h.z * w_inv};
}
// auxiliary.h:97-118
__device__ __forceinline__ f2 project_vec_approx(f3 p, f3 v, float tx, float ty)
{
    return {(v.x - v.z * p.x / p.z) / (p.z * tx), (v.y - v.z * p.y / p.z) / (p.z * ty)};
}
// auxiliary.h:128-139
__device__ __forceinline__ f2 dnormvdv(f2 v, f2 dv)
{
    float sum2 = v.x * v.x + v.y * v.y;
    float normv = sqrtf(sum2);
    float invsum32 = 1.0f / (normv * normv * normv);
    return {((sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y) * invsum32,
            (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y) * invsum32};
}
// auxiliary.h:141-152
__device__ __forceinline__ f3 dnormvdv(f3 v, f3 dv)
{
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float normv = sqrtf(sum2);
    float invsum32 = 1.0f / (normv * normv * normv);
    return {((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32,
            (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32,
            (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32};
}

// SH constants, auxiliary.h:11-26
__device__ constexpr float SH_C0 = 0.28209479177387814f;
__device__ constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                           SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
__device__ constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                           SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                           SH_C3_6 = -0.5900435899266435f;
} // namespace ts
