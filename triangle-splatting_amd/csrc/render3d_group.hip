// render3d_group.hip -- blend kernels of the 3D variant (rasterizer_type="3D"), lane-group edition.
//
// Behaviour follows FORWARD::renderCUDA / BACKWARD::renderCUDA of the reference's submodules/diff-triangle-rasterization-3D
// ("R3D": src/forward.cu:151-306, src/backward.cu:216-454): for every pixel the view-space ray
// p_ray = (tan_fovx * pixToProj(x), tan_fovy * pixToProj(y), 1) is intersected with the triangle's plane, barycentrics are taken
// in 3D, and the same ecc / alpha / front-to-back blend as in the 2D variant follows.  Kept quirks: the normal is unnormalised;
// accum_normal has no background term; the BACKWARD skip test is on G = exp(power), not on alpha (R3D backward.cu:351 vs
// forward.cu:265), so pairs with G >= 1/255 > alpha that the forward skipped still receive (tiny) gradients.
//
// Structure = render_group.hip (four 16-lane groups per wave, one triangle per 4x4 pixel block and step, per-group entry lists,
// DPP-row transpose-reduce, conflict-aware LDS accumulation, coalesced 64-byte atomic flush).  What is specific here:
//   * the per-pixel arithmetic is the REFERENCE'S, expression for expression: depth = v1.n / p_ray.n, p_view = depth p_ray,
//     p_vk = v_k - p_view, a1 = cross(p_v2, p_v3).n / n.n, a2 = cross(p_v3, p_v1).n / n.n, and the gradient terms of
//     backward.cu:376-420 -- round 1 used projective affine forms N_k(q) / Den(q) and moment sums, which deliberately left
//     the reference's rounding behaviour (its parity tests needed explained-outlier clauses);
//   * the lists are read in DENSE batches (round 5, as render_group.hip since round 4): the emission kernel marks in the top four bits of an
//     instance's value which quadrants the triangle's support -- scaled for the backward's G >= 1/255 test, projected -- can reach
//     (ts2d_support.h: quad_setup_3d), a quadrant wave gathers and culls only those entries, compacted by stream_refill;
//   * only CULLING uses the affine forms (N_k, Den affine in the in-quadrant pixel offset; a_k >= m  <=>  s (N_k - m Den) >= 0
//     wherever Den keeps its sign s over the 4x4 block), with the rounding slack added to the acceptance margin.
#include "ts2d_common.h"
#include "ts2d_wave.h"
#include "ts2d_group.h"
#include "ts2d_support.h"

#ifndef TS3G_BWD_WAVES // resident waves per SIMD the backward's register budget is declared for (occupancy experiments: tools/build_variant.sh)
#define TS3G_BWD_WAVES 6 // round 5: 80 registers with 3-4 spilled dwords outside the step loop; 1.128 vs 1.141 ms at the headline, 0.142 vs 0.148 at 93 k (profiles/r05_emission_variants.txt)
#endif
namespace
{
// Row of the constants table (ROW = 20 floats):
//   [0..3] v1.xyz v2.x   [4..7] v2.yz v3.xy   [8..11] v3.z n.xyz   [12..15] d0 = v1.n, 1/n.n, opacity, r   [16] g [17] b
//   [18] backward: triangle id; forward: this wave's running contrib_sum   [19] forward: running contrib_max
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vscale(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Cull3
{
    float d0, inn;
    bool ov[4];
};

// Conservative culling against the four 4x4 blocks of the quadrant.  With u2 = v2 - v1, u3 = v3 - v1 and r = hit point - v1,
// Den r = n x (delta x v1) where p_ray = v1 / v1.z + delta (see the derivation in render3d.hip: affine in the pixel offset, no
// pole at grazing incidence); a2 = r.(u3 x n) / n.n, a3 = r.(n x u2) / n.n, a1 = 1 - a2 - a3.
template <bool GAMMA1>
__device__ __forceinline__ Cull3 cull3(V3 v1, V3 v2, V3 v3, V3 n, float op, float g2, V3 ray0, float sx, float sy)
{
    Cull3 c;
    c.inn = 1.0f / vdot(n, n); // forward.cu:250
    c.d0 = vdot(v1, n);        // forward.cu:243
    const float dc = vdot(ray0, n), dx = n.x * sx, dy = n.y * sy; // Den(q) = dc + dx qx + dy qy
    const float iz = 1.0f / v1.z;
    const float ddx = ray0.x - v1.x * iz, ddy = ray0.y - v1.y * iz;
    const V3 m0 = vcross(n, V3{ddy * v1.z, -ddx * v1.z, ddx * v1.y - ddy * v1.x});
    const V3 mx = vcross(n, V3{0.0f, -sx * v1.z, sx * v1.y});
    const V3 my = vcross(n, V3{sy * v1.z, 0.0f, -sy * v1.x});
    const V3 G2 = vscale(c.inn, vcross(vsub(v3, v1), n)), G3 = vscale(c.inn, vcross(n, vsub(v2, v1)));
    const float a2c = vdot(m0, G2), a2x = vdot(mx, G2), a2y = vdot(my, G2);
    const float a3c = vdot(m0, G3), a3x = vdot(mx, G3), a3y = vdot(my, G3);
    const float a1c = dc - a2c - a3c, a1x = dx - a2x - a3x, a1y = dy - a2y - a3y;
    const float t = 255.0f * op;
    float E = -1.0f;
    if (t >= 1.0f) // alpha >= 1/255 (forward) / G >= 1/255 (backward, op <= 1) needs ecc <= E
    {
        const float L = 2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(t); // ecc^(2 gamma) <= 2 ln(255 op); the backward passes op = 1
        if (GAMMA1) E = __builtin_amdgcn_sqrtf(L);
        else E = (g2 < 1e-6f) ? 10.0f : pow_nonneg(L, 1.0f / g2);
        E = fminf(E * 1.0005f + 0.002f, 10.01f);
    }
    const float m = (1.0f - E) * (1.0f / 3.0f);
    // f_k(q) = N_k(q) - m Den(q), affine: coefficients (x, y, c)
    const float f1x = a1x - m * dx, f1y = a1y - m * dy, f1c = a1c - m * dc;
    const float f2x = a2x - m * dx, f2y = a2y - m * dy, f2c = a2c - m * dc;
    const float f3x = a3x - m * dx, f3y = a3y - m * dy, f3c = a3c - m * dc;
#pragma unroll
    for (int g = 0; g < 4; g++)
    {
        const float bx = (g & 1) ? 4.0f : 0.0f, by = (g >> 1) ? 4.0f : 0.0f;
        const float d00 = dc + dx * bx + dy * by;
        const float dmin = d00 + fminf(0.0f, 3.0f * dx) + fminf(0.0f, 3.0f * dy), dmax = d00 + fmaxf(0.0f, 3.0f * dx) + fmaxf(0.0f, 3.0f * dy);
        bool ov = E > 0.0f;
        if (dmin > 0.0f || dmax < 0.0f) // Den keeps its sign over the block
        {
            const float s = dmin > 0.0f ? 1.0f : -1.0f;
            const float slack = 1e-3f * fmaxf(fabsf(dmin), fabsf(dmax));
            const float h1 = s * (f1c + f1x * bx + f1y * by) + fmaxf(0.0f, 3.0f * s * f1x) + fmaxf(0.0f, 3.0f * s * f1y);
            const float h2 = s * (f2c + f2x * bx + f2y * by) + fmaxf(0.0f, 3.0f * s * f2x) + fmaxf(0.0f, 3.0f * s * f2y);
            const float h3 = s * (f3c + f3x * bx + f3y * by) + fmaxf(0.0f, 3.0f * s * f3x) + fmaxf(0.0f, 3.0f * s * f3y);
            ov = ov && h1 >= -slack && h2 >= -slack && h3 >= -slack;
        }
#ifdef TS3G_NO_CULL
        ov = true;
#endif
        c.ov[g] = ov;
    }
    return c;
}

// [19] = the entry's position in the tile's list; a list entry is the LDS byte offset of its row (render_group.hip, round 3)
// backward row: constants, k1 = n x (v3 - v2) and k2 = n x (v1 - v3) (the ray-independent halves of d a1 / d depth = n . ((v3 - v2) x p_ray)
// = p_ray . k1 and d a2 / d depth = p_ray . k2, backward.cu:389, 395: two dot products per pixel instead of two differences and two cross
// products), the entry's 16 gradient sums
constexpr int KROW3 = 8, SOFF3 = ROW + KROW3, BROW3 = SOFF3 + 16;
__device__ __forceinline__ void publish_k3(float *row, V3 v1, V3 v2, V3 v3, V3 n)
{
    const V3 k1 = vcross(n, vsub(v3, v2)), k2 = vcross(n, vsub(v1, v3));
    float4 *q = (float4 *)(row + ROW);
    q[0] = make_float4(k1.x, k1.y, k1.z, k2.x);
    q[1] = make_float4(k2.y, k2.z, 0.0f, 0.0f);
}
__device__ __forceinline__ void publish_row3(float *row, V3 v1, V3 v2, V3 v3, V3 n, const Cull3 &c, const float4 &r3, float w18, int jpos)
{
    float4 *q = (float4 *)row;
    q[0] = make_float4(v1.x, v1.y, v1.z, v2.x);
    q[1] = make_float4(v2.y, v2.z, v3.x, v3.y);
    q[2] = make_float4(v3.z, n.x, n.y, n.z);
    q[3] = make_float4(c.d0, c.inn, r3.x, r3.y);
    q[4] = make_float4(r3.z, r3.w, w18, __int_as_float(jpos));
}
// Second pass of a batch with more than NR surviving entries (rare): the lane gathers its entry's record again (see render_group.hip)
__device__ __forceinline__ void republish_row3(float *row, const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec, uint32_t pos,
                                               bool with_id, int jpos)
{
    const uint32_t id = point_list[pos] & TS_ID_MASK; // the top bits are the instance's quadrant mask
    const float4 *rp = rec + 4 * (size_t)id;
    const float4 r0 = rp[0], r1 = rp[1], r2 = rp[2], r3 = rp[3];
    const V3 v1 = {r0.x, r0.y, r0.z}, v2 = {r0.w, r1.x, r1.y}, v3 = {r1.z, r1.w, r2.x}, n = {r2.y, r2.z, r2.w};
    Cull3 c;
    c.inn = 1.0f / vdot(n, n);
    c.d0 = vdot(v1, n);
    publish_row3(row, v1, v2, v3, n, c, r3, with_id ? __uint_as_float(id) : 0.0f, jpos);
    if (with_id) publish_k3(row, v1, v2, v3, n); // backward rows
}
// Row -1: a unit triangle in the plane z = 1, a thousand units off axis, opacity 0: every pixel sees ecc ~ 3000
__device__ __forceinline__ void write_dummy_row3(float *row, int lane)
{
    if (lane < ROW)
    {
        // v1 = (1000, 1000, 1), v2 = (1001, 1000, 1), v3 = (1000, 1001, 1), n = (0, 0, 1), d0 = 1, 1/n.n = 1
        const float tab[ROW] = {1000.0f, 1000.0f, 1.0f, 1001.0f, 1000.0f, 1.0f, 1000.0f, 1001.0f, 1.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        row[lane] = lane == 19 ? __int_as_float(0x7fffffff) : tab[lane]; // list position of the dummy: beyond every pixel's range
    }
}

// The reference's per-pixel geometry (R3D forward.cu:238-256, backward.cu:328-343)
struct Hit3
{
    V3 n, p1, p2, p3, c1, c2; // normal, p_vk = v_k - p_view, c1 = cross(p_v2, p_v3), c2 = cross(p_v3, p_v1)
    float prn, inv_prn, depth, inn, a1, a2, a3, mn, ecc, op, r, g, b;
    bool ok; // |p_ray . n| >= EPS
};
// BWD selects which of the reference's two roundings of the depth is reproduced: the forward divides (forward.cu:243), the backward
// multiplies by the rounded reciprocal (backward.cu:330-331).  The hit point has the magnitude of the depth while the barycentrics
// live on the scale of the triangle, so one ulp of the depth is ~depth / edge ulps of a_k: v_rcp_f32 alone (1 ulp) tripled the
// distance to the reference's gradients (profiles/r02_noise_floor3d_93k.json); one Newton step restores correct rounding.
template <bool BWD>
__device__ __forceinline__ Hit3 hit3(const float *row, V3 ray)
{
    const float4 q0 = *(const float4 *)(row), q1 = *(const float4 *)(row + 4), q2 = *(const float4 *)(row + 8), q3 = *(const float4 *)(row + 12);
    Hit3 h;
    const V3 v1 = {q0.x, q0.y, q0.z}, v2 = {q0.w, q1.x, q1.y}, v3 = {q1.z, q1.w, q2.x};
    h.n = {q2.y, q2.z, q2.w};
    h.inn = q3.y; h.op = q3.z; h.r = q3.w;
    h.prn = vdot(ray, h.n);
    h.ok = fabsf(h.prn) >= 1e-8f; // forward.cu:241
    // a ray inside the plane is skipped by the reference; here its lanes run on with depth = 0 so that every value stays finite
    // (their alpha is forced to 0, and 0 * finite == 0 keeps them out of all sums)
    const float r0 = h.ok ? __builtin_amdgcn_rcpf(h.prn) : 0.0f;
    if (BWD)
    {
        h.inv_prn = fmaf(fmaf(-h.prn, r0, 1.0f), r0, r0);
        h.depth = q3.x * h.inv_prn;
    }
    else
    {
        const float q = q3.x * r0;
        h.inv_prn = r0;
        h.depth = fmaf(fmaf(-h.prn, q, q3.x), r0, q);
    }
    const V3 pv = vscale(h.depth, ray);
    h.p1 = vsub(v1, pv); h.p2 = vsub(v2, pv); h.p3 = vsub(v3, pv);
    h.c1 = vcross(h.p2, h.p3);
    h.c2 = vcross(h.p3, h.p1);
    h.a1 = vdot(h.c1, h.n) * h.inn; // forward.cu:251-253
    h.a2 = vdot(h.c2, h.n) * h.inn;
    h.a3 = 1.0f - h.a1 - h.a2;
    h.mn = fminf(fminf(h.a1, h.a2), h.a3);
    h.ecc = fmaf(-3.0f, h.mn, 1.0f);
    return h;
}

template <bool RICH, bool GAMMA1>
__global__ void __launch_bounds__(256, 6) render3d_fwd_group_kernel(RenderArgs a, float tan_fovx, float tan_fovy, const uint2 *__restrict__ ranges,
                                                                  const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
                                                                  float *__restrict__ final_T, uint32_t *__restrict__ n_contrib,
                                                                  float *__restrict__ out_feature, float *__restrict__ out_depth,
                                                                  float *__restrict__ out_normal, float *__restrict__ contrib_sum,
                                                                  float *__restrict__ contrib_max)
{
    __shared__ __attribute__((aligned(16))) float cst_all[4][(NR + 1) * ROW];
    __shared__ __attribute__((aligned(16))) uint32_t list_all[4][4 * NR / 2]; // per group: NR entries of (row | batch position << 8)
    constexpr int TCAP = 1024; // see render_group.hip: the tile's contribution statistics, merged over the four quadrant waves
    __shared__ unsigned long long tsum[RICH ? TCAP : 1]; // 16.48 fixed point (ts2d_group.h)
    __shared__ int tmax[RICH ? TCAP : 1];

    const int tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, sub = lane & 15;
    const int X0 = tx * TS_TILE + (wave & 1) * 8, Y0 = ty * TS_TILE + (wave >> 1) * 8;
    const int lx = ((grp & 1) << 2) + (sub & 3), ly = ((grp >> 1) << 2) + (sub >> 2);
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    // pixToProj(v, S) = (2 v - S + 1) / S   (R3D auxiliary.h:40-43)
    const V3 ray = {tan_fovx * ((2.0f * (float)px - (float)a.W + 1.0f) / (float)a.W), tan_fovy * ((2.0f * (float)py - (float)a.H + 1.0f) / (float)a.H), 1.0f};
    const float sx = tan_fovx * 2.0f / (float)a.W, sy = tan_fovy * 2.0f / (float)a.H;
    const V3 ray0 = {tan_fovx * ((2.0f * (float)X0 - (float)a.W + 1.0f) / (float)a.W), tan_fovy * ((2.0f * (float)Y0 - (float)a.H + 1.0f) / (float)a.H), 1.0f};
    const uint2 range = ranges[tile];
    const int len = (int)(range.y - range.x);
    if (RICH)
    {
        for (int k = threadIdx.x; k < min(len, TCAP); k += 256) { tsum[k] = 0ull; tmax[k] = 0; }
        __syncthreads();
    }
    const float g2 = 2.0f * a.gamma;
    const float bg0 = a.background[0], bg1 = a.C > 1 ? a.background[1] : 0.0f, bg2 = a.C > 2 ? a.background[2] : 0.0f;
    float *cst = cst_all[wave] + ROW;
    uint32_t *list = list_all[wave];
    write_dummy_row3(cst - ROW, lane);
    const char *lds0 = (const char *)cst_all;
    const uint32_t row0 = (uint32_t)(wave * (NR + 1) + 1) * (ROW * 4), dummy = row0 - ROW * 4;
    const int stat_step = ((lane >> 3) & 1) | ((lane >> 1) & 2) | ((lane << 1) & 4); // the step of a window whose statistics this lane ends up with

    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f, anx = 0.0f, any_ = 0.0f, anz = 0.0f, ad = 0.0f;
    bool done = !inside;
    uint32_t last = (uint32_t)len;

    // dense batches: only the entries whose quadrant bit is set are gathered and culled (ts2d_group.h, stream_refill); `pos` = list position
    uint32_t id = 0;
    int pos = 0, cursor = 0;
    for (;;)
    {
        const unsigned long long alive = ballot(!done);
        if (alive == 0) break;
        int nq = 0;
        stream_refill<false>(id, pos, nq, point_list + range.x, cursor, len, TS_ID_BITS + wave, lane);
        if (nq == 0) break;
        const bool valid = lane < nq;
        float4 r0 = make_float4(0, 0, 1, 0), r1 = make_float4(1, 0, 0, 1), r2 = make_float4(1, 0, 0, 1), r3 = make_float4(0, 0, 0, 0);
        if (valid)
        {
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
        }
        const V3 v1 = {r0.x, r0.y, r0.z}, v2 = {r0.w, r1.x, r1.y}, v3 = {r1.z, r1.w, r2.x}, n = {r2.y, r2.z, r2.w};
        const Cull3 c = cull3<GAMMA1>(v1, v2, v3, n, r3.x, g2, ray0, sx, sy);
        unsigned long long M[4];
#pragma unroll
        for (int g = 0; g < 4; g++) M[g] = ((alive >> (16 * g)) & 0xFFFFull) ? ballot(valid && c.ov[g] && r3.x * 255.0f >= 1.0f) : 0ull;
        const unsigned long long any = M[0] | M[1] | M[2] | M[3];
        if (any == 0) continue;
        // compacted table rows, at most NR per pass (render_group.hip)
        const bool anybit = (any >> lane) & 1;
        const int rank = lane_rank(any), nact = __popcll(any);
        const int r = rank & (NR - 1);
        bool mine = anybit && rank < NR;
        if (mine) publish_row3(cst + r * ROW, v1, v2, v3, n, c, r3, 0.0f, pos);
        for (int h = 0;;)
        {
            const unsigned long long mm = nact <= NR ? any : ballot(mine);
            list[lane] = dummy | (dummy << 16);
            int steps = 0;
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                const unsigned long long Mh = M[g] & mm;
                if ((Mh >> lane) & 1) ((u16a *)list)[g * NR + lane_rank(Mh)] = (unsigned short)(row0 + r * (ROW * 4));
                steps = max(steps, __popcll(Mh));
            }
            const uint32_t *mylist = list + grp * (NR / 2);
            for (int t0 = 0; t0 < steps; t0 += 8)
            {
                float cw[8];
                const uint4 packed = *(const uint4 *)(mylist + (t0 >> 1));
#pragma unroll
                for (int st = 0; st < 8; st++)
                {
                    cw[st] = 0.0f;
                    if (t0 + st < steps)
                    {
                        const uint32_t word = st < 2 ? packed.x : (st < 4 ? packed.y : (st < 6 ? packed.z : packed.w));
                        const float *row = (const float *)(lds0 + ((st & 1) ? (word >> 16) : (word & 0xFFFFu)));
                        const int jpos = __float_as_int(row[19]); // position in the tile's list
                        const Hit3 h = hit3<false>(row, ray);
                        const float pw = GAMMA1 ? h.ecc * h.ecc : pow_nonneg(h.ecc, g2);
                        const float alpha = fminf(0.99f, h.op * __builtin_amdgcn_exp2f(pw * -0.7213475204444817f)); // forward.cu:259-260
                        const bool hit = !done && h.ok && ecc_in_range(h.ecc) && alpha >= 1.0f / 255.0f;            // forward.cu:241,256,261
                        const float al = hit ? alpha : 0.0f;
                        const float contrib = al * T;
                        ar = fmaf(h.r, contrib, ar);
                        ag = fmaf(row[16], contrib, ag);
                        ab = fmaf(row[17], contrib, ab);
                        if (RICH)
                        {
                            anx = fmaf(h.n.x, contrib, anx); // forward.cu:276 (unnormalised normal)
                            any_ = fmaf(h.n.y, contrib, any_);
                            anz = fmaf(h.n.z, contrib, anz);
                            ad = fmaf(h.depth, contrib, ad); // forward.cu:277
                            cw[st] = contrib;
                        }
                        T *= (1.0f - al);
                        const bool sat = hit && T <= 0.0001f; // forward.cu:280
                        last = sat ? (uint32_t)(jpos + 1) : last;
                        done = done || sat;
                    }
                }
                if (RICH)
                {
                    // contrib_sum / contrib_max (forward.cu:271-273): reduced per 16-lane group, added to the tile's statistics in LDS
                    // with integer atomics (ts2d_group.h)
                    float sm, mx;
                    row_reduce8_sum_max(cw, 0xCCCCCCCCCCCCCCCCull, sm, mx);
                    int k = 0; // the list position of "its" step, from the step's row (render_group.hip)
                    if ((lane & 1) == 0 && sm > 0.0f) k = __float_as_int(*(const float *)(lds0 + ((const u16a *)list)[grp * NR + t0 + stat_step] + 19 * 4));
                    if ((lane & 1) == 0 && sm > 0.0f) tile_stats_add<TCAP>(tsum, tmax, k, sm, mx, point_list + range.x, contrib_sum, contrib_max);
                }
            }
            if (++h * NR >= nact) break;
            mine = anybit && rank >= NR;
            if (mine) republish_row3(cst + r * ROW, point_list, rec, range.x + pos, false, pos);
        }
    }
    // the wave's pixels leave first; their stores and the ids of the flush are in flight while the wave waits for the others (render_group.hip)
    if (inside)
    {
        const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_feature[pix] = ar + T * bg0;
        if (a.C > 1) out_feature[HW + pix] = ag + T * bg1;
        if (a.C > 2) out_feature[2 * HW + pix] = ab + T * bg2;
        if (RICH)
        {
            out_depth[pix] = ad + T * (a.background_depth_dev ? *a.background_depth_dev : a.background_depth);
            out_normal[pix] = anx;
            out_normal[HW + pix] = any_;
            out_normal[2 * HW + pix] = anz;
        }
    }
    if (RICH)
    {
        constexpr int NF = (TCAP + 255) / 256;
        const int nflush = min(len, TCAP);
        uint32_t ids[NF];
#pragma unroll
        for (int j = 0; j < NF; j++)
        {
            const int k = (int)threadIdx.x + 256 * j;
            ids[j] = k < nflush ? point_list[range.x + k] & TS_ID_MASK : 0u;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NF; j++)
        {
            const int k = (int)threadIdx.x + 256 * j;
            if (k < nflush)
            {
                const unsigned long long fx48 = tsum[k];
                if (fx48 != 0ull) tile_stats_flush(fx48, tmax[k], ids[j], contrib_sum, contrib_max);
            }
        }
    }
}

// Backward (R3D backward.cu:216-454).  Gradient record of the 3D variant: [0..8] dL/dv1_view dL/dv2_view dL/dv3_view,
// [9..11] dL/dnormal_view, [12] dL/dopacity, [13..15] dL/drgb.  Per pair, with (z1, z2, z3) = dL/da (only the arg-min component is
// non-zero) and w1 = z1 - z3, w2 = z2 - z3 (da3 = -da1 - da2 throughout, :396-400):
//   dL/dv2 = w1 cross(p_v3, n) / n.n                                  (:386, 404)
//   dL/dv3 = (w1 cross(n, p_v2) + w2 cross(p_v1, n)) / n.n            (:387, 393, 405)
//   dL/dv1 = w2 cross(n, p_v3) / n.n + dL_ddepth n / (p_ray.n)        (:391, 402-403)
//   dL_ddepth = dL_ddepth_pixel contrib + w1 da1_ddepth + w2 da2_ddepth, da1_ddepth = n.cross(v3 - v2, p_ray) / n.n, ...   (:389, 395, 401)
//   dL/dn  = dL_dnormal_pixel contrib + (w1 (c1 - 2 a1 n) + w2 (c2 - 2 a2 n)) / n.n + dL_ddepth p_v1 / (p_ray.n)   (:388, 394, 403, 406)
template <bool RICH, bool GAMMA1, int WPB> // WPB = quadrant waves per workgroup (1: single-wave workgroups, see render_group.hip)
__global__ void __launch_bounds__(64 * WPB, TS3G_BWD_WAVES) render3d_bwd_group_kernel(RenderArgs a, float tan_fovx, float tan_fovy, const uint2 *__restrict__ ranges,
                                                                     const uint32_t *__restrict__ point_list, const float4 *__restrict__ rec,
                                                                     const float *__restrict__ final_T, const uint32_t *__restrict__ n_contrib,
                                                                     const float *__restrict__ dL_dout_feature,
                                                                     const float *__restrict__ dL_dout_depth,
                                                                     const float *__restrict__ dL_dout_normal, float *__restrict__ grad_rec)
{
    __shared__ __attribute__((aligned(16))) float rows_all[WPB][(NR + 1) * BROW3];
    __shared__ __attribute__((aligned(16))) uint32_t list_all[WPB][4 * NR / 2];

    int tile, quad, wave;
    if (WPB == 4)
    {
        tile = tile_of_block(blockIdx.x, a.grid_x, a.grid_y);
        quad = wave = threadIdx.x >> 6;
    }
    else
    {
        // single-wave workgroups: four consecutive units of an XCD are the four quadrants of one tile (they stay neighbours in dispatch order
        // and on one XCD: shared L2 for the tile's list and records)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = tile_of_block(((j >> 2) << 3) | x, a.grid_x, a.grid_y);
        quad = j & 3;
        wave = 0;
    }
    if (tile < 0) return; // the grid is padded (ts2d_wave.h)
    const int tx = tile % a.grid_x, ty = tile / a.grid_x;
    const int lane = threadIdx.x & 63;
    const int grp = lane >> 4, sub = lane & 15;
    const int X0 = tx * TS_TILE + (quad & 1) * 8, Y0 = ty * TS_TILE + (quad >> 1) * 8;
    const int lx = ((grp & 1) << 2) + (sub & 3), ly = ((grp >> 1) << 2) + (sub >> 2);
    const int px = X0 + lx, py = Y0 + ly;
    const bool inside = px < a.W && py < a.H;
    const V3 ray = {tan_fovx * ((2.0f * (float)px - (float)a.W + 1.0f) / (float)a.W), tan_fovy * ((2.0f * (float)py - (float)a.H + 1.0f) / (float)a.H), 1.0f};
    const float sx = tan_fovx * 2.0f / (float)a.W, sy = tan_fovy * 2.0f / (float)a.H;
    const V3 ray0 = {tan_fovx * ((2.0f * (float)X0 - (float)a.W + 1.0f) / (float)a.W), tan_fovy * ((2.0f * (float)Y0 - (float)a.H + 1.0f) / (float)a.H), 1.0f};
    const uint2 range = ranges[tile];
    const float g2 = 2.0f * a.gamma;
    const size_t pix = (size_t)py * a.W + px, HW = (size_t)a.H * a.W;
    float *rows = rows_all[wave] + BROW3;
    uint32_t *list = list_all[wave];
    write_dummy_row3(rows - BROW3, lane);
    if (lane < KROW3) (rows - BROW3)[ROW + lane] = 0.0f; // k1, k2 of the dummy row
    char *lds0 = (char *)rows_all;
    const uint32_t row0 = (uint32_t)(wave * (NR + 1) + 1) * (BROW3 * 4), dummy = row0 - BROW3 * 4;
    const uint32_t accoff = SOFF3 * 4 + 4 * sub;

    float T = inside ? final_T[pix] : 0.0f;
    const int last = inside ? (int)n_contrib[pix] : 0;
    float dpr = 0.0f, dpg = 0.0f, dpb = 0.0f, dnx = 0.0f, dny = 0.0f, dnz = 0.0f, dd = 0.0f, B = 0.0f;
    if (inside) // backward.cu:283-295; one scalar back-to-front composite B = sum_c dL_dpix_c accum_c (see render.hip)
    {
        dpr = dL_dout_feature[pix];
        B = dpr * a.background[0];
        if (a.C > 1) { dpg = dL_dout_feature[HW + pix]; B = fmaf(dpg, a.background[1], B); }
        if (a.C > 2) { dpb = dL_dout_feature[2 * HW + pix]; B = fmaf(dpb, a.background[2], B); }
        if (RICH)
        {
            dnx = dL_dout_normal[pix]; dny = dL_dout_normal[HW + pix]; dnz = dL_dout_normal[2 * HW + pix];
            dd = dL_dout_depth[pix];
            B = fmaf(dd, a.background_depth_dev ? *a.background_depth_dev : a.background_depth, B);
        }
    }
    float lm = (float)last;
    lm = fmaxf(lm, dpp<DPP_XOR1>(lm));
    lm = fmaxf(lm, dpp<DPP_XOR2>(lm));
    lm = fmaxf(lm, dpp<DPP_HALF_MIRROR>(lm));
    lm = fmaxf(lm, dpp<DPP_MIRROR>(lm));
    int glast[4];
#pragma unroll
    for (int g = 0; g < 4; g++) glast[g] = (int)__builtin_amdgcn_readlane((int)lm, 16 * g);
    const int maxlast = max(max(glast[0], glast[1]), max(glast[2], glast[3]));
    if (maxlast <= 0) return;

    // dense batches, walked back to front: lane 0 holds the entry farthest back (ts2d_group.h, stream_refill<true>); `pos` = list position
    uint32_t id = 0;
    int pos = 0, cursor = maxlast;
    for (;;)
    {
        int nq = 0;
        stream_refill<true>(id, pos, nq, point_list + range.x, cursor, maxlast, TS_ID_BITS + quad, lane);
        if (nq == 0) break;
        const bool valid = lane < nq;
        float4 r0 = make_float4(0, 0, 1, 0), r1 = make_float4(1, 0, 0, 1), r2 = make_float4(1, 0, 0, 1), r3 = make_float4(0, 0, 0, 0);
        if (valid)
        {
            const float4 *rp = rec + 4 * (size_t)id;
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
        }
        const V3 v1 = {r0.x, r0.y, r0.z}, v2 = {r0.w, r1.x, r1.y}, v3 = {r1.z, r1.w, r2.x}, n = {r2.y, r2.z, r2.w};
        // the backward's skip test is on G (backward.cu:351): support computed with opacity 1
        const Cull3 c = cull3<GAMMA1>(v1, v2, v3, n, 1.0f, g2, ray0, sx, sy);
        unsigned long long M[4];
#pragma unroll
        for (int g = 0; g < 4; g++) M[g] = ballot(valid && c.ov[g] && pos < glast[g]); // entries at or behind glast[g] are skipped by all of block g's pixels
        const unsigned long long any = M[0] | M[1] | M[2] | M[3];
        if (any == 0) continue;
        const bool anybit = (any >> lane) & 1;
        const int rank = lane_rank(any), nact = __popcll(any);
        const int r = rank & (NR - 1);
        bool mine = anybit && rank < NR; // back to front = the low lanes first
        if (mine)
        {
            publish_row3(rows + r * BROW3, v1, v2, v3, n, c, r3, __uint_as_float(id), pos);
            publish_k3(rows + r * BROW3, v1, v2, v3, n);
        }
        for (int h = (nact - 1) / NR;;)
        {
            const unsigned long long mm = nact <= NR ? any : ballot(mine);
            if (mine)
            {
                float4 *z = (float4 *)(rows + r * BROW3 + SOFF3);
                z[0] = z[1] = z[2] = z[3] = make_float4(0, 0, 0, 0);
            }
            list[lane] = dummy | (dummy << 16);
            int steps = 0;
#pragma unroll
            for (int g = 0; g < 4; g++)
            {
                const unsigned long long Mh = M[g] & mm;
                const int nn = __popcll(Mh);
                if ((Mh >> lane) & 1) ((u16a *)list)[g * NR + lane_rank(Mh)] = (unsigned short)(row0 + r * (BROW3 * 4));
                steps = max(steps, nn);
            }
            const u16a *mylist = (const u16a *)list + grp * NR;
            unsigned long long conflict;
            {
                const u16a *l16 = (const u16a *)list + (lane & (NR - 1));
                const uint32_t l0 = l16[0], l1 = l16[NR], l2 = l16[2 * NR], l3 = l16[3 * NR];
                conflict = ballot(lane < NR && ((l0 != dummy && (l0 == l1 || l0 == l2 || l0 == l3)) || (l1 != dummy && (l1 == l2 || l1 == l3)) ||
                                                (l2 != dummy && l2 == l3)));
            }
            uint32_t ra_next = mylist[0];
            for (int t0 = 0; t0 < steps; t0++)
            {
                const uint32_t ra = ra_next;
                ra_next = mylist[min(t0 + 1, NR - 1)]; // one step ahead (render_group.hip)
                const float *row = (const float *)(lds0 + ra);
                float *acc = (float *)(lds0 + ra + accoff);
                const int jpos = __float_as_int(row[19]);
                const bool shared_row = (conflict >> t0) & 1;
                const float acc0 = *acc;
                const Hit3 h = hit3<true>(row, ray);
                const float cg = row[16], cb = row[17];
                const float pw = GAMMA1 ? h.ecc * h.ecc : pow_nonneg(h.ecc, g2);
                const float G = __builtin_amdgcn_exp2f(pw * -0.7213475204444817f);
                const float opG = h.op * G;
                const float alpha = fminf(0.99f, opG);
                const bool hit = (jpos < last) && h.ok && ecc_in_range(h.ecc) && G >= 1.0f / 255.0f; // backward.cu:322-323,328,343,351
                const float al = hit ? alpha : 0.0f;
                const float oma = 1.0f - al;
                T = T * __builtin_amdgcn_rcpf(oma); // :354
                const float contrib = al * T;
                float X = fmaf(dpb, cb, fmaf(dpg, cg, dpr * h.r)); // :368
                if (RICH) // :374-380
                {
                    X = fmaf(dnz, h.n.z, fmaf(dny, h.n.y, fmaf(dnx, h.n.x, X)));
                    X = fmaf(dd, h.depth, X);
                }
                const float dL_dcontrib = X - B;
                B = fmaf(al, X, oma * B);
                const float dL_dalpha = dL_dcontrib * T; // :383
                // -3 dL_decc, :384-385 (gamma = 1: pw / (ecc + 1e-8) is ecc to 1e-8 / ecc relative, see render_group.hip)
                const float zr = GAMMA1 ? 1.5f * g2 * (dL_dalpha * alpha) * h.ecc : 1.5f * g2 * (dL_dalpha * alpha) * pw * __builtin_amdgcn_rcpf(h.ecc + 1e-8f);
                const float z = (hit && opG < 0.99f) ? zr : 0.0f;
                const bool k1 = h.a1 == h.mn;
                const bool k2 = !k1 && h.a2 == h.mn;
                const float z1 = k1 ? z : 0.0f, z2 = k2 ? z : 0.0f, z3 = z - z1 - z2;
                const float w1 = (z1 - z3) * h.inn, w2 = (z2 - z3) * h.inn; // 1 / n.n folded in
                const V3 A1 = vcross(h.p3, h.n), A2 = vcross(h.n, h.p2), A3 = vcross(h.p1, h.n);
                const float4 k0 = *(const float4 *)(row + ROW);
                const float2 k4 = *(const float2 *)(row + ROW + 4);
                const float da1_dd = vdot(ray, V3{k0.x, k0.y, k0.z}), da2_dd = vdot(ray, V3{k0.w, k4.x, k4.y}); // :389, 395 (see KROW3)
                const float dLdd = fmaf(w2, da2_dd, fmaf(w1, da1_dd, RICH ? dd * contrib : 0.0f)) * h.inv_prn; // dL_ddepth / (p_ray.n)
                float v[16];
                v[bitrev4(0)] = fmaf(dLdd, h.n.x, -w2 * A1.x); // dL/dv1 = w2 cross(n, p_v3) + dL_ddepth n / prn; cross(n, p_v3) = -A1
                v[bitrev4(1)] = fmaf(dLdd, h.n.y, -w2 * A1.y);
                v[bitrev4(2)] = fmaf(dLdd, h.n.z, -w2 * A1.z);
                v[bitrev4(3)] = w1 * A1.x; v[bitrev4(4)] = w1 * A1.y; v[bitrev4(5)] = w1 * A1.z;
                v[bitrev4(6)] = fmaf(w2, A3.x, w1 * A2.x); v[bitrev4(7)] = fmaf(w2, A3.y, w1 * A2.y); v[bitrev4(8)] = fmaf(w2, A3.z, w1 * A2.z);
                const float s12 = -2.0f * (w1 * h.a1 + w2 * h.a2);
                v[bitrev4(9)] = fmaf(dLdd, h.p1.x, fmaf(s12, h.n.x, fmaf(w2, h.c2.x, fmaf(w1, h.c1.x, RICH ? dnx * contrib : 0.0f))));
                v[bitrev4(10)] = fmaf(dLdd, h.p1.y, fmaf(s12, h.n.y, fmaf(w2, h.c2.y, fmaf(w1, h.c1.y, RICH ? dny * contrib : 0.0f))));
                v[bitrev4(11)] = fmaf(dLdd, h.p1.z, fmaf(s12, h.n.z, fmaf(w2, h.c2.z, fmaf(w1, h.c1.z, RICH ? dnz * contrib : 0.0f))));
                v[bitrev4(12)] = hit ? dL_dalpha * G : 0.0f; // :447
                v[bitrev4(13)] = dpr * contrib; v[bitrev4(14)] = dpg * contrib; v[bitrev4(15)] = dpb * contrib; // :365
                const float red = row_reduce16(v, 0xCCCCCCCCCCCCCCCCull, 0xAAAAAAAAAAAAAAAAull);
                if (!shared_row) *acc = acc0 + red;
                else
                {
#pragma unroll
                    for (int g = 0; g < 4; g++)
                    {
                        if (grp == g) *acc += red;
                        wave_lds_order();
                    }
                }
            }
            {
                const int nn = __popcll(mm);
#pragma unroll 1
                for (int e0 = 0; e0 < nn; e0 += 4)
                {
                    const int e = e0 + grp;
                    if (e < nn)
                    {
                        const uint32_t eid = __float_as_uint(rows[e * BROW3 + 18]);
                        unsafeAtomicAdd(grad_rec + TS_GRAD_FLOATS * (size_t)eid + sub, rows[e * BROW3 + SOFF3 + sub]);
                    }
                }
            }
            if (--h < 0) break;
            mine = anybit && rank >= NR;
            if (mine) republish_row3(rows + r * BROW3, point_list, rec, range.x + pos, true, pos);
        }
    }
}
} // namespace

#define TS_DISPATCH_G3(KERNEL, ...)                                                                                                \
    do                                                                                                                              \
    {                                                                                                                               \
        const bool g1 = (a.gamma == 1.0f);                                                                                          \
        if (a.rich_info && g1) hipLaunchKernelGGL((KERNEL<true, true>), grid, dim3(256), 0, s, __VA_ARGS__);                        \
        else if (a.rich_info) hipLaunchKernelGGL((KERNEL<true, false>), grid, dim3(256), 0, s, __VA_ARGS__);                        \
        else if (g1) hipLaunchKernelGGL((KERNEL<false, true>), grid, dim3(256), 0, s, __VA_ARGS__);                                 \
        else hipLaunchKernelGGL((KERNEL<false, false>), grid, dim3(256), 0, s, __VA_ARGS__);                                        \
    } while (0)

void ts_launch_render3d_fwd_group(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g, const BinningStateView &b,
                                  const ImageStateView &im, float *out_feature, float *out_depth, float *out_normal, float *contrib_sum,
                                  float *contrib_max, hipStream_t s)
{
    if (a.grid_x * a.grid_y == 0) return;
    const dim3 grid((unsigned)ts_tile_units(a.grid_x, a.grid_y));
    TS_DISPATCH_G3(render3d_fwd_group_kernel, a, tan_fovx, tan_fovy, im.ranges, b.vals, g.rec, im.final_T, im.n_contrib, out_feature, out_depth,
                   out_normal, contrib_sum, contrib_max);
}

void ts_launch_render3d_bwd_group(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g, const BinningStateView &b,
                                  const ImageStateView &im, const float *dL_dout_feature, const float *dL_dout_depth,
                                  const float *dL_dout_normal, float *grad_rec, hipStream_t s)
{
    const dim3 grid((unsigned)(a.grid_x * a.grid_y));
    if (grid.x == 0) return;
    constexpr int WPB = 1;
    const dim3 grid1((unsigned)((WPB == 4 ? 1 : 4) * ts_tile_units(a.grid_x, a.grid_y))); // padded: units past the image return at once
    const bool g1 = (a.gamma == 1.0f);
#define TS_BWD3(R, G) hipLaunchKernelGGL((render3d_bwd_group_kernel<R, G, WPB>), grid1, dim3(64 * WPB), 0, s, a, tan_fovx, tan_fovy, im.ranges, b.vals, \
                                         g.rec, im.final_T, im.n_contrib, dL_dout_feature, dL_dout_depth, dL_dout_normal, grad_rec)
    if (a.rich_info && g1) TS_BWD3(true, true);
    else if (a.rich_info) TS_BWD3(true, false);
    else if (g1) TS_BWD3(false, true);
    else TS_BWD3(false, false);
#undef TS_BWD3
}
