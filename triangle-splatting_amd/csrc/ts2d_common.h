// ts2d_common.h -- private layout of the three opaque state buffers and kernel launch prototypes.
//
// The reference bump-allocates SoA arrays inside torch uint8 tensors (R2D/src/param_struct.h:11-125).
// The layout here is free to differ (the buffers are opaque to callers) and is designed for MI355X:
//   * one 64-byte "render record" per triangle (a full 64 B sector of a 128 B HBM/L2 line) so that the
//     per-instance gather in the blend kernels is exactly four dwordx4 loads from one line;
//   * one 64-byte gradient record per triangle so that a wave's reduced gradients land with a single
//     16-lane global_atomic_add_f32 on one line;
//   * tile rectangles packed to 8 bytes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define TS_TILE 16
#define TS_ALIGN 256

// Render record (16 floats = 64 B), written by preprocess_fwd, read by render_fwd / render_bwd.
//   [0..5]  v1.x v1.y v2.x v2.y v3.x v3.y   screen-space vertices        (reference: GeometryState.v{1,2,3}_2D)
//   [6]     opacity                          copied from the input         (reference: gathered separately)
//   [7..9]  r g b                            SH colour or feature          (reference: GeometryState.rgb / feature)
//   [10..12] normal_view.xyz                 rich_info only                (reference: GeometryState.normal_view)
//   [13..15] v_depth.xyz                     rich_info only                (reference: GeometryState.v_depth)
// area2 is not stored: it is recomputed as cross(v2-v1, v3-v1), the expression the reference stores.
#define TS_REC_FLOATS 16

// Quadrant masks (ts2d_support.h): the emission kernel marks in the four spare bits of an instance's value which 8x8 quadrants of its tile the
// triangle's support can reach, and the blend kernels' quadrant waves gather and cull only those entries, in dense batches.  Round 4,
// measured against plain lists alternating on one box (profiles/r04_qmask.txt): render_fwd 0.429 -> 0.398 ms, render_bwd 0.839 -> 0.791,
// emission 0.029 -> 0.053, step 1.658 -> 1.604.  The plain-list front end was deleted from the product in round 5 (it lives on in the lab
// kernels render.hip / render_q8.hip, which ignore the mask bits): triangle ids are < 2^28 everywhere (validate() in api.hip).
#define TS_ID_BITS 28
#define TS_ID_MASK 0x0FFFFFFFu
// Gradient record (16 floats = 64 B), accumulated by render_bwd, consumed by preprocess_bwd.
//   [0..5] dL/dv{1,2,3}_2D   [6] dL/dopacity   [7..9] dL/drgb   [10..12] dL/dnormal_view   [13..15] dL/dv_depth
#define TS_GRAD_FLOATS 16

// Scratch of one stable LSD radix pass over n (key, value) pairs (binning.hip): per-chunk digit counts, their per-slab
// and per-digit prefixes.  One chunk = consecutive pairs handled by one workgroup; one slab = 64 chunks.  The chunk length is a property
// of the sort, picked from its size: 4096 pairs for instance lists of millions (longer digit runs = better coalesced scatter stores; keys and
// values through one staging array), 1024 for everything up to 2.5 M pairs -- such a sort is a few hundred to a few thousand workgroups that
// are all resident at once, and each launch lasts as long as ONE workgroup's chain of ranking steps (16 / 8 / 4 per lane at 4096 / 2048 /
// 1024): depth sort of 1 M keys 45 -> 40 us against 2048 (round 3: 2048 against 4096 0.103 vs 0.113 ms), of 93 k keys 36 -> 33, tile sort of
// 1 M instances 35 -> 31 (profiles/r05_small_chunks_ab.txt) -- and 2048 for depth sorts beyond that (5 M triangles: 39 slabs, still ticket-free).
#ifndef TS_RS_CHUNK
#define TS_RS_CHUNK 4096
#endif
#define TS_RS_CHUNK_MID 2048
#define TS_RS_CHUNK_SMALL 1024
#define TS_RS_SMALL_BELOW 2500000 /* pairs */
// Chunk length of the instance sort, from the CAPACITY of the binning state (so that the forward and the backward, which both carve from the
// buffer's size, agree); the tables are sized for the shortest chunks at every capacity, so that bytes(capacity) grows with the capacity.
static inline int ts_instance_chunk(size_t n) { return n <= (size_t)TS_RS_SMALL_BELOW ? TS_RS_CHUNK_SMALL : TS_RS_CHUNK; }
static inline int ts_depth_chunk(size_t n) { return n <= (size_t)TS_RS_SMALL_BELOW ? TS_RS_CHUNK_SMALL : TS_RS_CHUNK_MID; }
#define TS_RS_BINS 256
#define TS_RS_TICKET_EXTRA 8 /* words behind the per-slab tickets; [slabs + 4] = the depth sort's top_const flag.  (The census of the depth sort's
                                first histogram needs no words here: its per-chunk sums / key-bit ORs / ANDs borrow g.blocksum, g.tiles_sorted and
                                g.offsets, which the scan rewrites afterwards -- binning.hip, ts_sort_by_depth_begin.) */
struct RadixScratchView
{
    uint32_t *table;   // chunks x 256   count of digit d in chunk c, then (in place) its exclusive prefix inside the slab
    uint32_t *slabtot; // slabs x 256    per-slab totals, then (in place) their exclusive prefix over the slabs
    uint32_t *binbase; // 256            exclusive prefix of the digit totals
    uint32_t *tickets; // TS_RS_TICKETS  "last block finishes" tickets: [0] pass, [1] scan blocks, [2 + slab] per slab; zero between launches
    uint32_t *slabacc[2]; // slabs x 256 each: per-slab digit totals of the ticket-free passes (binning.hip, rs_hist_direct_kernel), accumulated
                          // with atomics; pass p uses [p & 1], and whoever runs before it has cleared that buffer
    int chunks, slabs;
    int chunk; // pairs per chunk: TS_RS_CHUNK, TS_RS_CHUNK_MID or TS_RS_CHUNK_SMALL
};

struct GeometryStateView
{
    float4 *rec;             // P * 4 float4
    float *depth;            // P   sort key (centroid view-space z; its bit pattern is the radix key)
    uint32_t *tiles_touched; // P
    uint2 *rect;             // P   x = minx | miny << 16, y = maxx | maxy << 16
    uint8_t *clamped;        // P   bit c set when colour channel c was clamped at 0
    uint32_t *sk[2], *sv[2]; // P each: ping-pong (key, value) buffers of the depth sort
    uint32_t *depth_sorted;  //     = sk[1]: depth keys in ascending order (after the 4th pass)
    uint32_t *perm;          //     = sv[1]: triangle ids in (depth, id) order after the 4th pass
    uint32_t *top_const;     // device flag (one of the sort's scratch words, cleared with the tickets): the visible triangles' depth keys share
                             // their top byte, the 4th pass was skipped and the order is in sk[0] / sv[0] (binning.hip, sorted_ids())
    uint32_t *tiles_sorted;  // P   tiles_touched[perm[i]]
    uint32_t *offsets;       // P   inclusive prefix sum of tiles_sorted: instance slots of the i-th nearest triangle
    uint64_t *blocksum;      // ceil(P / 1024) + 2   raw per-block sums of tiles_sorted; [nblocks] = N (scratch of the depth sort's census before that)
    uint64_t *supersum;      // ceil(nblocks / 64) + 1   sums of 64 consecutive block sums (atomics; zeroed by the step's first launch)
    RadixScratchView rs;
};

struct BinningStateView
{
    uint32_t *k[2], *v[2];   // N each: ping-pong (tile id, triangle id) buffers; instances are emitted into k[0] / v[0]
    uint32_t *tile;          //     = k[passes & 1]: sorted by tile (stable => depth order inside a tile)
    uint32_t *vals;          //     = v[passes & 1]: triangle id per sorted instance
    int passes;              // radix passes of 8 bits needed for the tile count
    RadixScratchView rs;
};

struct ImageStateView
{
    uint2 *ranges;       // T   [start, end) of each tile in the sorted list
    uint32_t *n_contrib; // W*H
    float *final_T;      // W*H
    int32_t *status;     // 4   [0] sync-free forward: 1 = the instance count exceeded the capacity of the binning state
};

static inline size_t ts_align_up(size_t v) { return (v + TS_ALIGN - 1) & ~(size_t)(TS_ALIGN - 1); }

template <typename T>
static inline void ts_carve(char *&p, T *&out, size_t count)
{
    p = (char *)ts_align_up((size_t)p);
    out = (T *)p;
    p += count * sizeof(T);
}

// `reserve_chunk`: the chunk length the tables are SIZED for (<= chunk) -- the instance sort picks its chunk length from the capacity, and the
// bytes of a binning state must grow with the capacity (ts_binning_capacity inverts them), so its tables always have the short chunks' size.
static inline void ts_carve_radix(char *&p, size_t n, RadixScratchView &r, int chunk = TS_RS_CHUNK, int reserve_chunk = 0)
{
    r.chunk = chunk;
    r.chunks = (int)((n + chunk - 1) / chunk);
    r.slabs = (r.chunks + 63) / 64;
    const int rc = reserve_chunk > 0 ? reserve_chunk : chunk;
    const size_t chunks = (n + rc - 1) / rc, slabs = (chunks + 63) / 64;
    ts_carve(p, r.table, chunks * TS_RS_BINS);
    ts_carve(p, r.slabtot, slabs * TS_RS_BINS);
    ts_carve(p, r.binbase, (size_t)TS_RS_BINS);
    ts_carve(p, r.tickets, slabs + TS_RS_TICKET_EXTRA);
    ts_carve(p, r.slabacc[0], slabs * TS_RS_BINS);
    ts_carve(p, r.slabacc[1], slabs * TS_RS_BINS);
}

static inline size_t ts_carve_geometry(char *base, int32_t P, GeometryStateView &v)
{
    char *p = base;
    size_t n = (size_t)(P > 0 ? P : 0);
    ts_carve(p, v.rec, n * 4);
    ts_carve(p, v.depth, n);
    ts_carve(p, v.tiles_touched, n);
    ts_carve(p, v.rect, n);
    ts_carve(p, v.clamped, n);
    ts_carve(p, v.sk[0], n);
    ts_carve(p, v.sk[1], n);
    ts_carve(p, v.sv[0], n);
    ts_carve(p, v.sv[1], n);
    v.depth_sorted = v.sk[1]; // four 8-bit passes: depth -> sk[0] -> sk[1] -> sk[0] -> sk[1]
    v.perm = v.sv[1];
    ts_carve(p, v.tiles_sorted, n);
    ts_carve(p, v.offsets, n);
    ts_carve(p, v.blocksum, (n + 1023) / 1024 + 2);
    ts_carve(p, v.supersum, ((n + 1023) / 1024 + 63) / 64 + 1);
    ts_carve_radix(p, n, v.rs, ts_depth_chunk(n));
    v.top_const = v.rs.tickets + v.rs.slabs + 4;
    return (size_t)(p - base) + TS_ALIGN;
}

static inline int ts_higher_msb(uint32_t n) // R2D/src/rasterizer.cu:20-35 (same result: bits needed above n's msb search)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1)
    {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return (int)msb;
}

// Key bits of the instance sort: the tile ids are 0 .. ntiles - 1.  (The reference sorts 32 + getHigherMsb(ntiles) bits, rasterizer.cu:211-222 -- one
// more than needed when ntiles is a power of two, e.g. 9 for the 256 tiles of a 256 x 256 image: here that is one 8-bit pass instead of two.)
static inline int ts_tile_bits(int ntiles) { return ts_higher_msb((uint32_t)(ntiles > 1 ? ntiles - 1 : 1)); }

static inline size_t ts_carve_binning(char *base, int64_t N, int32_t W, int32_t H, BinningStateView &v)
{
    char *p = base;
    size_t n = (size_t)(N > 0 ? N : 0);
    int gx = (W + TS_TILE - 1) / TS_TILE, gy = (H + TS_TILE - 1) / TS_TILE;
    ts_carve(p, v.k[0], n);
    ts_carve(p, v.k[1], n);
    ts_carve(p, v.v[0], n);
    ts_carve(p, v.v[1], n);
    v.passes = (ts_tile_bits(gx * gy) + 7) / 8; // tile bits only (see binning.hip)
    v.tile = v.k[v.passes & 1];
    v.vals = v.v[v.passes & 1];
    ts_carve_radix(p, n, v.rs, ts_instance_chunk(n), TS_RS_CHUNK_SMALL);
    return (size_t)(p - base) + TS_ALIGN;
}

// The capacity (in tile instances) of a binning buffer of `bytes` bytes: the largest N whose carving fits.  The binning state is ALWAYS carved
// for this capacity, so the forward that filled a buffer and the backward that reads it agree on the layout from the buffer's size alone --
// whatever instance count the one or the other was told (the exact count of the reference's sequence, or less than the capacity of a
// speculative / sync-free forward).
static inline int64_t ts_binning_capacity(size_t bytes, int32_t W, int32_t H)
{
    BinningStateView v;
    if (bytes < ts_carve_binning(nullptr, 0, W, H, v)) return -1; // not even an empty state fits
    int64_t lo = 0, hi = (int64_t)(bytes / 16) + 1;                // four u32 arrays per instance: bytes(N) >= 16 N
    if (hi > 0x7fffffffll) hi = 0x7fffffffll;
    while (lo < hi)
    {
        const int64_t mid = lo + (hi - lo + 1) / 2;
        if (ts_carve_binning(nullptr, mid, W, H, v) <= bytes) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Narrows the radix scratch of a binning state carved for a larger capacity to the `n` instances the host knows are there (pointers stay where the
// capacity put them; every launch of the forward must see the same view).
static inline void ts_binning_set_count(BinningStateView &v, int64_t n)
{
    v.rs.chunks = (int)((n + v.rs.chunk - 1) / v.rs.chunk);
    v.rs.slabs = (v.rs.chunks + 63) / 64;
}

static inline size_t ts_carve_image(char *base, int32_t W, int32_t H, ImageStateView &v)
{
    char *p = base;
    int gx = (W + TS_TILE - 1) / TS_TILE, gy = (H + TS_TILE - 1) / TS_TILE;
    ts_carve(p, v.ranges, (size_t)gx * gy);
    ts_carve(p, v.n_contrib, (size_t)W * H);
    ts_carve(p, v.final_T, (size_t)W * H);
    ts_carve(p, v.status, (size_t)4);
    return (size_t)(p - base) + TS_ALIGN;
}

// ---- kernel launchers (defined in the .hip files) -------------------------------------------------
struct PreprocessArgs
{
    int W, H, P, D, M, C;
    int grid_x, grid_y;
    bool rich_info, use_shs, back_culling;
    float tan_fovx, tan_fovy;
    const float *viewmatrix, *projmatrix, *campos;
    const float *vertex, *shs, *feature, *opacity;
};

void ts_launch_preprocess_fwd(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s, int mode = 0); // mode: ts2d_preprocess_launch.h (PRE_ALL / PRE_NOCOLOUR)
bool ts_preprocess_fwd_splittable(const PreprocessArgs &a);
void ts_launch_preprocess_colour(const PreprocessArgs &a, const GeometryStateView &g, int variant, int blocks, hipStream_t s); // the SH colours behind a PRE_NOCOLOUR launch
// binning.hip -- every step hand-written for gfx950 (the round-1 rocPRIM calls survive only as test comparators)
void ts_sort_by_depth_begin(const GeometryStateView &g, int32_t P, unsigned long long *host_out, hipStream_t s); // first histogram + N + key-bit census
void ts_sort_by_depth_finish(const GeometryStateView &g, int32_t P, hipStream_t s);          // the rest: (depth bits, id) -> sorted ids
void ts_scan_offsets(const GeometryStateView &g, int32_t P, hipStream_t s);                  // tiles_sorted, block sums + their groups' sums
// How the emission kernel forms the quadrant masks in the values' top bits (ts2d_support.h).  2D: the screen triangle of the render record, its
// support scaled by E(opacity, 2 gamma).  3D: the view-space triangle scaled by E(opacity = 1) -- the 3D backward's skip test is on G, not on
// alpha (R3D backward.cu:351) -- about its centroid IN ITS PLANE, projected to pixels; a ray meets the plane inside the scaled triangle exactly
// where the pixel lies inside that projection, so the 2D test with E = 1 on the projected triangle is the 3D test.
struct QuadMaskArgs
{
    int variant;    // 2 or 3
    float g2;       // 2 gamma (< 1e-6: the whole ecc <= 10 region)
    float tan_fovx, tan_fovy;
    int W, H;
    float inv_W, inv_H; // 1 / W, 1 / H (the 3D setup's pixel -> ray conversions; the masks' margins absorb their rounding)
};
void ts_launch_zero_words(uint32_t *p, size_t n, hipStream_t s); // binning.hip
void ts_launch_emit_keys(int P, int grid_x, int ntiles, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                         float *contrib_sum, float *contrib_max, int64_t capacity, int32_t *status, const QuadMaskArgs &qm, hipStream_t s); // offsets + instances (+ output clears); capacity < 0: synchronous path
const unsigned long long *ts_instance_count_dev(const GeometryStateView &g, int P);                        // where the scan leaves N
void ts_sort_pairs(const BinningStateView &b, int64_t N, const unsigned long long *n_dev, int ntiles, hipStream_t s); // stable, tile bits only
void ts_launch_tile_ranges(int64_t N, const unsigned long long *n_dev, const BinningStateView &b, const ImageStateView &im, hipStream_t s);
size_t ts_quantile_scratch_bytes();                                                               // select.hip: torch.quantile of non-negative floats by radix select
void ts_quantile_threshold(const uint32_t *keys, size_t n, float q, void *scratch, float *thr, hipStream_t s);
void ts_quantile_passes(const uint32_t *keys, size_t n, float q, void *scratch, int first_pass, hipStream_t s); // for callers that weave the select into their own kernels (ts2d_select.h)
size_t ts_quantile_state_words();
size_t ts_radix_scratch_bytes(size_t n);                                                          // the same sort for other callers (knn.hip)
int ts_radix_sort_pairs(uint32_t *const k[2], uint32_t *const v[2], size_t n, int end_bit, void *scratch, hipStream_t s, bool force_tickets = false);
void ts_force_ticket_passes(bool on); // lab library only (csrc/ts2d_lab.h): no exported entry point of the product library reaches it
void ts_force_depth_pass4(bool on);   // likewise
void ts_lab_depth_split(int mode, int bucket_cap); // mode 1: never the sampled-splitter depth order, 2: up to 1.6 M triangles; bucket_cap > 0: its per-bucket register capacity

struct RenderArgs
{
    int W, H, C, grid_x, grid_y;
    float gamma, background_depth;
    const float *background; // C floats, device
    const float *background_depth_dev; // optional: one float on the device that overrides background_depth (ts2d_geometry)
    bool rich_info;
    int ablate; // profiling only (env TS2D_ABLATE, builds with -DTS2D_ABLATION): 0 = full kernel; see render.hip
    int legacy_blend; // measurement only (env TS2D_BLEND=wave, read once): round-1 one-triangle-per-wave blend kernels
    int bwd_mfma;  // experiment (env TS2D_BWD=mfma): render_bwd forms its per-entry sums with f32 MFMA instead of VALU reduction networks
};
void ts_launch_render_fwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                          const ImageStateView &im, float *out_feature, float *out_depth, float *out_normal,
                          float *contrib_sum, float *contrib_max, hipStream_t s);
void ts_launch_render_bwd(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                          const ImageStateView &im, const float *dL_dout_feature, const float *dL_dout_depth,
                          const float *dL_dout_normal, float *grad_rec, hipStream_t s);
// lane-group blend kernels (render_group.hip): four 4x4 pixel blocks per wave, one triangle per block and step
void ts_launch_render_fwd_group(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                                const ImageStateView &im, float *out_feature, float *out_depth, float *out_normal,
                                float *contrib_sum, float *contrib_max, hipStream_t s);
void ts_launch_render_bwd_group(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b,
                                const ImageStateView &im, const float *dL_dout_feature, const float *dL_dout_depth,
                                const float *dL_dout_normal, float *grad_rec, hipStream_t s);
// queue kernels (render_q8.hip; lab library only, selected with TS2D_BLEND=q8): eight 4x2 pixel blocks per wave, each walking its own queue of triangles
void ts_launch_render_fwd_q8(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                             float *out_feature, float *out_depth, float *out_normal, float *contrib_sum, float *contrib_max, hipStream_t s);
void ts_launch_render_bwd_q8(const RenderArgs &a, const GeometryStateView &g, const BinningStateView &b, const ImageStateView &im,
                             const float *dL_dout_feature, const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec, hipStream_t s);
void ts_launch_preprocess_bwd(const PreprocessArgs &a, const int32_t *radii, const GeometryStateView &g,
                              const float *grad_rec, float *dL_dvertex, float *dL_dcenter2D, float *dL_dshs,
                              float *dL_dfeature, float *dL_dopacity, hipStream_t s);

// ---- 3D variant (TS2D_FLAG_3D): same states and binning, its own record contents and blend maths -----------------
// Render record: [0..8] v1_view v2_view v3_view   [9..11] normal_view (unnormalised)   [12] opacity   [13..15] r g b
// Gradient record: [0..8] dL/dv{1,2,3}_view   [9..11] dL/dnormal_view   [12] dL/dopacity   [13..15] dL/drgb
void ts_launch_preprocess3d_fwd(const PreprocessArgs &a, int32_t *radii, const GeometryStateView &g, hipStream_t s, int mode = 0);
void ts_launch_preprocess3d_bwd(const PreprocessArgs &a, const int32_t *radii, const GeometryStateView &g,
                                const float *grad_rec, float *dL_dvertex, float *dL_dcenter2D, float *dL_dshs,
                                float *dL_dfeature, float *dL_dopacity, hipStream_t s);
void ts_launch_render3d_fwd(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                            const BinningStateView &b, const ImageStateView &im, float *out_feature, float *out_depth,
                            float *out_normal, float *contrib_sum, float *contrib_max, hipStream_t s);
void ts_launch_render3d_bwd(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                            const BinningStateView &b, const ImageStateView &im, const float *dL_dout_feature,
                            const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec, hipStream_t s);

// lane-group kernels with the reference's per-pixel ray / plane arithmetic (render3d_group.hip); the default
void ts_launch_render3d_fwd_group(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                                  const BinningStateView &b, const ImageStateView &im, float *out_feature, float *out_depth,
                                  float *out_normal, float *contrib_sum, float *contrib_max, hipStream_t s);
void ts_launch_render3d_bwd_group(const RenderArgs &a, float tan_fovx, float tan_fovy, const GeometryStateView &g,
                                  const BinningStateView &b, const ImageStateView &im, const float *dL_dout_feature,
                                  const float *dL_dout_depth, const float *dL_dout_normal, float *grad_rec, hipStream_t s);

// ---- factored SH-gradient exchange (multi-GPU, shgrad.hip) ---------------------------------------------------------
void ts_launch_sh_grad_expand(int P, int D, int M, int V, const float *vertex, const float *campos, const float *dL_dcolor,
                              float *dL_dshs, hipStream_t s);

// ---- fused photometric loss (photometric.hip, include/ts_loss.h) ----------------------------------------------------
size_t ts_loss_workspace_bytes(int C, int H, int W);
hipError_t ts_loss_forward(const float *image, const float *gt, int C, int H, int W, float w_l1, float w_ssim, bool need_grad,
                           void *workspace, float *out, hipStream_t s);
hipError_t ts_loss_backward(const float *image, const float *gt, int C, int H, int W, float w_l1, float w_ssim, const void *workspace,
                            const float *grad_out, float *dL_dimage, hipStream_t s);

// ---- fused depth / normal consistency loss (depth_normal.hip, include/ts_loss.h) --------------------------------------------------
// aux_losses.hip -- DoGLoss / SmoothnessLoss (include/ts_loss.h)
size_t ts_aux_loss_workspace_bytes(int C, int H, int W, double scale);
hipError_t ts_dog_mask(const float *gt, int C, int H, int W, double sigma1, int ksize1, double sigma2, int ksize2, int invert, double scale, void *workspace,
                       float *mask, hipStream_t s);
hipError_t ts_smoothness_mask(const float *gt, int C, int H, int W, double scale, float quantile, void *workspace, float *mask, hipStream_t s);
hipError_t ts_masked_l1_forward(const float *img, const float *gt, const float *mask, int C, int H, int W, void *workspace, float *out, hipStream_t s);
hipError_t ts_masked_l1_backward(const float *img, const float *gt, const float *mask, int C, int H, int W, const float *grad_out, float *dimg, hipStream_t s);
hipError_t ts_scharr_smoothness_forward(const float *img, const float *mask, int C, int H, int W, void *workspace, float *out, hipStream_t s);
hipError_t ts_scharr_smoothness_backward(const float *img, const float *mask, int C, int H, int W, void *workspace, const float *grad_out, float *dimg,
                                         hipStream_t s);
hipError_t ts_downsample_forward(const float *in, int C, int H, int W, int h, int w, float *out, hipStream_t s);   // resample.hip
hipError_t ts_downsample_backward(const float *gout, int C, int H, int W, int h, int w, float *gin, hipStream_t s);
hipError_t ts_downsample_forward_planes(int n, const float *const *src, int H, int W, int h, int w, float *const *dst, hipStream_t s);
hipError_t ts_downsample_backward_planes(int n, const float *const *gout, int H, int W, int h, int w, float *const *gin, hipStream_t s);
size_t ts_depth_normal_workspace_bytes(int H, int W, double scale);
hipError_t ts_depth_normal_forward(const float *depth, const float *normal, int H, int W, float tan_fovx, float tan_fovy, double scale, float quantile,
                                   void *workspace, float *out, hipStream_t s);
hipError_t ts_depth_normal_backward(const float *depth, const float *normal, int H, int W, float tan_fovx, float tan_fovy, double scale,
                                    const void *workspace, const float *grad_out, float *dL_ddepth, float *dL_dnormal, hipStream_t s);

// ---- exact nearest-neighbour helpers (knn.hip, include/ts_knn.h) -------------------------------------------------------
size_t ts_knn_workspace_bytes(int P);
hipError_t ts_knn_mean_dist3(int P, const float *points, float *mean_dist2, void *ws, hipStream_t s);
hipError_t ts_knn_nearest_other(int P, int group, const float *points, uint32_t *nearest, void *ws, hipStream_t s);

// ---- fused Adam step (optim.hip, include/ts_optim.h) -------------------------------------------------------------------
struct tso_adam_slice;
struct tso_sh_factored_step;
hipError_t ts_optim_adam_step(const tso_adam_slice *slices, int n, double beta1, double beta2, double eps, hipStream_t s);
hipError_t ts_optim_adam_step_sh_factored(const tso_sh_factored_step &a, double beta1, double beta2, double eps, hipStream_t s);

// ---- per-iteration model-update statistics (model_update.hip, include/ts_model.h) --------------------------------------
hipError_t ts_model_training_statistic(int P, int V, const int32_t *radii, const float *c2d_grad, const float *csum, const float *cmax,
                                       float *g_accum, float *g_denom, float *max_radii, float *s_csum, float *s_cmax, float *c_denom,
                                       hipStream_t s);
size_t ts_model_select_scratch_bytes(int P);
hipError_t ts_model_select_rows(int P, const uint8_t *mask, int match, uint32_t *pos, uint32_t *scratch, uint32_t *count_host, hipStream_t s);
hipError_t ts_model_scatter_rows(int64_t rows, int row_words, const uint32_t *pos, const void *src, void *dst, int64_t dst_row0, hipStream_t s);
hipError_t ts_model_gather_rows(int64_t rows, int row_words, const uint32_t *idx, const void *src, void *dst, int64_t dst_row0, hipStream_t s);
hipError_t ts_model_grow_classify(int P, const float *vertex, float *g_accum, float *g_denom, float min_view_count, float grad_threshold,
                                  float split_scale_threshold, uint8_t *code, hipStream_t s);
hipError_t ts_model_split_vertex(int n_split, const uint32_t *parents, const float *vertex, float *child1, float *child2, hipStream_t s);
hipError_t ts_model_update_mask(int P, int mode, const float *opacity, const float *vertex, const float *max_radii, float a, float b, uint8_t *mask,
                                hipStream_t s);
hipError_t ts_model_clip(int P, int mode, const uint8_t *mask, float value, float *param, float *exp_avg, float *exp_avg_sq, hipStream_t s);
hipError_t ts_model_opacity_reset(int P, float reset_value, float *opacity, float *exp_avg, float *exp_avg_sq, hipStream_t s);
hipError_t ts_model_max_distance(int n_vertices, const float *vertex, const float *campos, float *out, hipStream_t s);
