/*
 * ts2d.h -- C ABI of the MI355X-native 2D differentiable triangle rasterizer (libts2d.so).
 *
 * This is the drop-in boundary for the hot path of GaodeRender/triangle-splatting's
 * `diff_triangle_rasterization_2D` extension ("R2D" = submodules/diff-triangle-rasterization-2D in
 * the reference tree).  Every entry point takes plain device pointers and sizes (no torch types);
 * the caller owns all memory, including the three opaque state buffers that the reference keeps in
 * torch uint8 tensors (geometryBuffer / binningBuffer / imageBuffer,
 * R2D/src/extension_interface.cu:126-128).  All kernels are enqueued on the HIP stream passed in
 * (`stream` is a hipStream_t; NULL = the null stream).
 *
 * What each entry point replaces in the reference:
 *
 *   ts2d_forward_bin      -> first half of Rasterizer::forward: preprocessCUDA + InclusiveSum + the
 *                            D2H read of num_rendered         (R2D/src/rasterizer.cu:116-193)
 *   ts2d_forward_render   -> second half: duplicateWithKeys + SortPairs + identifyTileRanges +
 *                            FORWARD::renderCUDA              (R2D/src/rasterizer.cu:195-266)
 *   ts2d_backward         -> Rasterizer::backward: BACKWARD::renderCUDA + BACKWARD::preprocessCUDA
 *                                                             (R2D/src/rasterizer.cu:269-358)
 *   ts2d_*_state_bytes    -> BaseDataBuffer::requiredSize     (R2D/src/param_struct.h:36-40)
 *   ts2d_forward_speculative -> the whole of Rasterizer::forward with its num_rendered read-back taken off the GPU's critical path
 *                            (R2D/src/rasterizer.cu:101-267; what the Python package calls)
 *
 * Together, forward_bin + forward_render are what `rasterizeTrianglesForward`
 * (R2D/src/extension_interface.cu:19-152, pybind name `rasterize_triangles`, R2D/ext.cpp:6) calls, and
 * ts2d_backward is what `rasterizeTrianglesBackward` (:154-260, `rasterize_triangles_backward`,
 * R2D/ext.cpp:8) calls.  The split of forward into two calls exists because the caller must size the
 * binning buffer from num_rendered, exactly like the reference's resize at rasterizer.cu:195.
 *
 * Return value: TS2D_OK (0) or a TS2D_ERR_* code; ts2d_last_error() returns a thread-local,
 * human-readable message for the last failing call.  Error behaviour mirrors the reference's
 * AT_ERROR checks (extension_interface.cu:53-81): bad shapes / C > 3 / gamma < 0 -> TS2D_ERR_INVALID.
 *
 * Matrix convention (unchanged from the reference): viewmatrix / projmatrix are 16 floats, element
 * [col*4+row] of the usual column-vector matrix, i.e. the row-major storage of the transposed
 * matrices built by src/diff_recon/utils/camera.py:112-114; see R2D/src/auxiliary.h:40-58.
 */
#ifndef TS2D_H
#define TS2D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TS2D_OK 0
#define TS2D_ERR_INVALID 1 /* argument validation failed (reference: AT_ERROR) */
#define TS2D_ERR_HIP 2     /* a HIP runtime call or kernel failed */
#define TS2D_ERR_CAPACITY 3 /* a caller-provided buffer is too small */

#define TS2D_FLAG_BACK_CULLING 0x1u /* R2D settings.back_culling */
#define TS2D_FLAG_RICH_INFO 0x2u    /* R2D settings.rich_info: depth, normal, contrib_sum/max + their grads */
#define TS2D_FLAG_DEBUG 0x4u        /* R2D settings.debug: synchronise + check after every kernel (auxiliary.h:358-367) */
#define TS2D_FLAG_USE_SHS 0x8u      /* colour from SH coefficients (extension_interface.cu:44) instead of `feature` */
#define TS2D_FLAG_3D 0x10u          /* rasterizer_type "3D": the ray/plane variant, submodules/diff-triangle-rasterization-3D
                                       (same entry points and argument lists, R3D/src/extension_interface.cu:14-276;
                                       R3D has no `contiguous` check, its forward.cu/backward.cu replace the 2D maths) */

#define TS2D_FLAG_SH_FACTORED 0x20u  /* ts2d_backward only, SH mode: do not write dL_dshs (may be null); dL_dfeature receives
                                       the clamp-masked colour gradient dL_dRGB (P*3) from which ts2d_sh_grad_expand
                                       rebuilds dL_dshs -- the multi-GPU exchange format (new; no reference counterpart) */

#define TS2D_MAX_CHANNELS 3 /* R2D/src/config.h:3 */
#define TS2D_TILE 16        /* R2D/src/config.h:4-5 (BLOCK_X = BLOCK_Y = 16) */

/* CameraInfo, R2D/src/param_struct.h:127-136.  All pointers are device pointers. */
typedef struct ts2d_camera
{
    int32_t width, height;
    float tan_fovx, tan_fovy;
    const float *viewmatrix; /* 16 floats */
    const float *projmatrix; /* 16 floats */
    const float *campos;     /* 3 floats */
} ts2d_camera;

/* GeometryInfo, R2D/src/param_struct.h:138-153.  All pointers are device pointers. */
typedef struct ts2d_geometry
{
    int32_t P;         /* number of triangles; at most 2^28 - 1 (the instance lists keep four bits of each value for a quadrant mask: TS2D_ERR_CAPACITY) */
    int32_t sh_degree; /* active SH degree D, 0..3 */
    int32_t M;         /* SH coefficients stored per triangle ((max_degree+1)^2); 0 in feature mode */
    int32_t C;         /* colour channels: 3 in SH mode, feature.size(1) <= 3 otherwise */
    float gamma;
    float scale_modifier; /* accepted and ignored, like the reference (param_struct.h:147) */
    float background_depth;
    const float *background; /* C floats */
    const float *vertex;     /* P*3*3 floats, world space */
    const float *shs;        /* P*M*3 floats (SH mode) or NULL */
    const float *feature;    /* P*C floats (feature mode) or NULL */
    const float *opacity;    /* P floats */
    const float *background_depth_dev; /* optional DEVICE pointer to one float: when non-NULL the kernels read the background depth from
                                          it and `background_depth` is ignored.  The reference's model computes it on the device every
                                          step (max |campos - vertex|, src/diff_recon/models/VanillaTS_model.py:623) and its binding turns
                                          the 0-dim tensor into a host float -- a full device synchronisation per forward (SURVEY.md 8a,
                                          row a1); handing over the pointer keeps the step asynchronous.  Same fp32 value either way. */
} ts2d_geometry;

/* ForwardOutput, R2D/src/param_struct.h:155-166.  depth/normal/contrib_* may be NULL without RICH_INFO.
 * Every element of every non-NULL output is written by the library (no pre-zeroing needed). */
typedef struct ts2d_forward_out
{
    float *out_feature;  /* C*H*W */
    float *depth;        /* H*W */
    float *normal;       /* 3*H*W */
    float *contrib_sum;  /* P */
    float *contrib_max;  /* P */
} ts2d_forward_out;

/* LossInput, R2D/src/param_struct.h:176-181. */
typedef struct ts2d_loss_grads
{
    const float *dL_dout_feature; /* C*H*W */
    const float *dL_dout_depth;   /* H*W   (RICH_INFO only) */
    const float *dL_dout_normal;  /* 3*H*W (RICH_INFO only).  With RICH_INFO, depth and normal may BOTH be NULL: "no gradient arrives on
                                   * them" = what two images of zeros give (the reference's autograd materialises those), by the colour-only
                                   * pixel kernel and without the two fills */
} ts2d_loss_grads;

/* BackwardOutput, R2D/src/param_struct.h:183-190.  Every element is written by the library. */
typedef struct ts2d_backward_out
{
    float *dL_dvertex;   /* P*3*3 */
    float *dL_dcenter2D; /* P*2 */
    float *dL_dshs;      /* P*M*3 (SH mode; may be NULL otherwise) */
    float *dL_dfeature;  /* P*C: dL/dfeature, or dL/d(rgb) in SH mode like the reference (rasterizer.cu:333) */
    float *dL_dopacity;  /* P */
} ts2d_backward_out;

/* The three opaque state buffers (private layout) + the backward scratch. */
typedef struct ts2d_state
{
    void *geometry; size_t geometry_bytes; /* >= ts2d_geometry_state_bytes(P) */
    void *binning;  size_t binning_bytes;  /* >= ts2d_binning_state_bytes(N, W, H) */
    void *image;    size_t image_bytes;    /* >= ts2d_image_state_bytes(W, H) */
} ts2d_state;

const char *ts2d_version(void);
const char *ts2d_last_error(void);

size_t ts2d_geometry_state_bytes(int32_t P);
size_t ts2d_binning_state_bytes(int64_t num_rendered, int32_t width, int32_t height);
size_t ts2d_image_state_bytes(int32_t width, int32_t height);
size_t ts2d_backward_scratch_bytes(int32_t P);
/* The instance capacity of a binning buffer of `bytes` bytes (the largest N with ts2d_binning_state_bytes(N, W, H) <= bytes).  The library
 * ALWAYS lays the binning state out for this capacity, so ts2d_backward finds the layout of the forward that filled the buffer from the
 * buffer's size alone: a buffer may be larger than the count needs (speculative and sync-free forwards), never smaller. */
int64_t ts2d_binning_capacity(size_t bytes, int32_t width, int32_t height);

/* Per-triangle preprocess, depth order and prefix sum.  Writes radii[P] and the geometry state and returns num_rendered (the
 * reference's blocking cudaMemcpy, rasterizer.cu:191).  The count is summed right after the per-triangle kernel and handed to the
 * host through a pinned word; the depth sort and the scan are queued behind it, so on return they may still be running on
 * `stream` -- the host waits for the count only, never for the whole stream. */
int ts2d_forward_bin(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii,
                     const ts2d_state *state, int64_t *num_rendered, void *stream);

/* Key emission, (tile, depth) sort, tile ranges and the per-pixel blend.  Fully asynchronous. */
int ts2d_forward_render(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t num_rendered,
                        const ts2d_state *state, const ts2d_forward_out *out, void *stream);

/* Backward.  `scratch` is >= ts2d_backward_scratch_bytes(P) bytes of device memory (zeroed by the
 * library).  Fully asynchronous. */
int ts2d_backward(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t num_rendered,
                  const int32_t *radii, const ts2d_state *state, const ts2d_loss_grads *loss, void *scratch,
                  size_t scratch_bytes, const ts2d_backward_out *out, void *stream);
/* The same backward with its last kernel (the per-triangle one, BACKWARD::preprocessCUDA's counterpart) launched over `num_ranges` (1..64)
 * consecutive triangle ranges -- range k = triangles [k * per, min(P, (k + 1) * per)), per = ceil(P / num_ranges) rounded up to a multiple of
 * 64 (ts2d_backward_range_rows) -- and, when `range_done_events` is not NULL, hipEventRecord(range_done_events[k], stream) behind range k: the
 * rows of that range in EVERY gradient output are final when the event fires.  Multi-GPU callers start the exchange of range k while range
 * k + 1 is computed.  Results are those of ts2d_backward (= num_ranges 1, no events). */
int ts2d_backward_ranged(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int64_t num_rendered, const int32_t *radii,
                         const ts2d_state *state, const ts2d_loss_grads *loss, void *scratch, size_t scratch_bytes,
                         const ts2d_backward_out *out, int32_t num_ranges, void *const *range_done_events, void *stream);
/* Rows per range of ts2d_backward_ranged for P triangles split into num_ranges. */
int32_t ts2d_backward_range_rows(int32_t P, int32_t num_ranges);

/* Sync-free forward (no counterpart in the reference, whose Rasterizer::forward blocks on a cudaMemcpy of num_rendered,
 * R2D/src/rasterizer.cu:191): preprocess, depth order, instance count, emission, tile sort, ranges and blend are enqueued in ONE call
 * and the host never waits.  The caller provides the capacity: `state->binning` must hold ts2d_binning_state_bytes(instance_capacity,
 * W, H) bytes, and `instance_capacity` is what it later passes to ts2d_backward as num_rendered (the binning state is carved for it).
 * The true instance count stays on the device; if it exceeds the capacity NOTHING is emitted (the image is the background, every
 * statistic zero) and the state's status word is set -- read it with ts2d_forward_status (one blocking read, e.g. once per step
 * after the optimizer has been queued) and re-run with a larger capacity.  Everything else as ts2d_forward_bin + ts2d_forward_render. */
int ts2d_forward(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii, const ts2d_state *state,
                 int64_t instance_capacity, const ts2d_forward_out *out, void *stream);
/* Speculative forward: the reference's interface (num_rendered comes back to the host, R2D/src/rasterizer.cu:189-191) without its stall.
 * Everything -- preprocess, depth order, count, emission, tile sort, ranges, blend -- is enqueued for the CAPACITY of the binning buffer the
 * caller guessed (ts2d_binning_capacity of state->binning_bytes; ts2d_instance_capacity_hint proposes one), and only then does the host
 * wait for the instance count, which the GPU publishes ~0.1 ms into the forward through a pinned word: the GPU never idles for the host's
 * round trip, the host is back while the blend kernel still runs, and *num_rendered is exact.  If *num_rendered exceeds the capacity the
 * speculative launches emitted nothing (the image is the background): the caller allocates ts2d_binning_state_bytes(*num_rendered, W, H)
 * and calls ts2d_forward_render(*num_rendered) -- the per-triangle half is done and valid.  state->binning may be NULL: then only
 * that half runs (== ts2d_forward_bin) and the caller ALWAYS follows with ts2d_forward_render.  A non-NULL buffer too small for a single
 * instance is refused (TS2D_ERR_CAPACITY): it would queue no render and a scene of zero instances would pass the overflow test.
 * ts2d_backward takes the exact *num_rendered. */
int ts2d_forward_speculative(const ts2d_camera *cam, const ts2d_geometry *geom, uint32_t flags, int32_t *radii, const ts2d_state *state,
                             const ts2d_forward_out *out, int64_t *num_rendered, void *stream);
/* A capacity for the next forward of P triangles at this image size on the current device: 1.25 x the recent maximum of instances per
 * triangle that forwards of the same variant (flags & TS2D_FLAG_3D) and size returned here, or 0 when there is no history yet. */
int64_t ts2d_instance_capacity_hint(int32_t P, int32_t width, int32_t height, uint32_t flags);
/* The histories behind the hint are kept per (device, variant, image size, KEY).  The key is the calling thread's (default 0): a caller with
 * several streams of views of one size -- train and evaluation cameras, two models in one process -- names them so that one does not size
 * the other's buffers.  ts2d_speculative_overflow_count: how many ts2d_forward_speculative calls of this process returned a num_rendered
 * above their capacity (each costs the caller a second render): a counter that keeps growing says the hint's 1.25 x margin is too small
 * for the caller's sequence of views (use a key per camera group, or the exact two-call form). */
void ts2d_set_capacity_hint_key(uint64_t key);
uint64_t ts2d_speculative_overflow_count(void);
/* overflowed = 1 when the last ts2d_forward on this state exceeded its capacity; num_rendered = the true instance count. */
int ts2d_forward_status(const ts2d_state *state, int32_t P, int32_t width, int32_t height, int32_t *overflowed, int64_t *num_rendered,
                        void *stream);

/* Multi-GPU gradient exchange helper (new capability, BASELINE.json north_star "image-parallel"; the reference has no
 * distributed path).  Each view's dL_dshs is basis(dir) x dL_dRGB per triangle (R2D/src/backward.cu:9-119), so ranks
 * exchange dL_dRGB (3 floats per triangle and view, from ts2d_backward with TS2D_FLAG_SH_FACTORED) plus the camera
 * centres instead of 3*M floats, and rebuild the sum over `num_views` views here:
 *   dL_dshs[i,k,:] = sum_v basis_k(normalize(centroid_i - campos[v])) * dL_dcolor[v,i,:]        (zeros for k >= (D+1)^2)
 * vertex: P*9, campos: num_views*3, dL_dcolor: num_views*P*3, dL_dshs: P*M*3 (fully written).  Asynchronous. */
int ts2d_sh_grad_expand(int32_t P, int32_t sh_degree, int32_t M, int32_t num_views, const float *vertex, const float *campos,
                        const float *dL_dcolor, float *dL_dshs, void *stream);

/* Timing hook used by bench.py: when enabled, ts2d_forward_render / ts2d_backward bracket each kernel
 * with HIP events on `stream`; ts2d_profile_read copies out (name, total_ms, launches) rows. */
void ts2d_profile_enable(int on);
void ts2d_profile_only(const char *kernel_name); /* NULL or "" = time every kernel; else only the named one */
void ts2d_profile_reset(void);
int ts2d_profile_read(int32_t index, char *name, size_t name_bytes, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* TS2D_H */
