/* ts_model.h -- C ABI of the model-update operators, exported by libts2d.so: the per-iteration statistics
 * and the building blocks of the periodic densification / pruning / clipping rules.
 *
 * Replaces the six boolean-mask statements of VanillaTSModel._training_statistic
 * (src/diff_recon/models/VanillaTS_model.py:347-363), which run after every training backward pass and consume the
 * rasterizer's radii / contrib_sum / contrib_max outputs and center2D.grad:
 *     visible = radii > 0                                        (VanillaTS_model.py:679, "visible_mask")
 *     gradient_accum[visible] += |center2D.grad[visible, :2]|    (:358)
 *     gradient_denom[visible] += 1                               (:359)
 *     contrib_sum[visible]   = max(contrib_sum[visible],  contrib_sum_view[visible])    (:360)
 *     contrib_max[visible]   = max(contrib_max[visible],  contrib_max_view[visible])    (:361)
 *     contrib_denom[visible] += 1                                (:362)
 *     max_radii2D[visible]   = max(max_radii2D[visible], radii[visible])                (:363)
 * Eager torch turns each line into nonzero + gather + op + scatter (with a host synchronisation per mask); here it is one
 * HBM-bound pass.  `num_views` > 1 applies the update for several views of the same step in one launch (image-parallel
 * training: ranks all-gather the four per-view arrays; every operation is order independent, so all ranks end up with
 * identical state).  All pointers are device pointers. */
#ifndef TS_MODEL_H
#define TS_MODEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-view inputs are laid out view-major: radii (V, P) int32; center2D_grad (V, P, 2); contrib_sum / contrib_max (V, P)
 * (may be NULL when the render ran without rich_info: those two running maxima are then left untouched).
 * State arrays are (P,) float32, updated in place. */
int tsm_training_statistic(int32_t P, int32_t num_views, const int32_t *radii, const float *center2D_grad, const float *contrib_sum,
                           const float *contrib_max, float *gradient_accum, float *gradient_denom, float *max_radii2D,
                           float *contrib_sum_state, float *contrib_max_state, float *contrib_denom, void *stream);

/* ---- periodic structural updates (VanillaTS_model.py:214-345 state surgery, :365-537 rules) ----------------------------------------
 * Building blocks; the host mirror diff_recon_hip/model_update.py strings them together under the reference's method names.
 * Rows are per-triangle records of `row_bytes` (a multiple of 4) in contiguous (P, ...) arrays: parameters, Adam moments, statistics. */

/* Stable compaction plan: pos[i] = number of rows j < i with mask[j] == match when mask[i] == match, else 0xFFFFFFFF; *count =
 * number of selected rows (one blocking read, like the reference's `.sum().item()`).  scratch: tsm_select_scratch_bytes(P). */
size_t tsm_select_scratch_bytes(int32_t P);
int tsm_select_rows(int32_t P, const uint8_t *mask, int32_t match, uint32_t *pos, void *scratch, size_t scratch_bytes, uint32_t *count,
                    void *stream);
/* dst[dst_row0 + pos[i]] = src[i] for every selected row i   (`param[mask]`, `exp_avg[mask]`, ... :224-227, 229-234) */
int tsm_scatter_rows(int64_t rows, int32_t row_bytes, const uint32_t *pos, const void *src, void *dst, int64_t dst_row0, void *stream);
/* dst[dst_row0 + j] = src[idx[j]], j < rows                  (clone / split attribute rows, :265-288) */
int tsm_gather_rows(int64_t rows, int32_t row_bytes, const uint32_t *idx, const void *src, void *dst, int64_t dst_row0, void *stream);
/* _densification's selection (:376-382) + _grow_points' clone / split classification (:261-263): code[i] = 0 untouched, 1 clone,
 * 2 split (mean side length > split_scale_threshold); gradient_accum / gradient_denom of the selected rows are reset to 0. */
int tsm_grow_classify(int32_t P, const float *vertex, float *gradient_accum, float *gradient_denom, float min_view_count,
                      float grad_threshold, float split_scale_threshold, uint8_t *code, void *stream);
/* the two children of each split triangle (:270-283): vertex rows of parents[j] cut at the centre of their longest side */
int tsm_split_vertex(int32_t n_split, const uint32_t *parents, const float *vertex, float *child1, float *child2, void *stream);
/* masks of the pruning / clipping rules: mode 0 sigmoid(opacity) < a (:391), 1 sigmoid(opacity) > a (:403),
 * 2 max_radii2D > a || mean side > b (:417-419), 3 mean side > a (:453) */
int tsm_update_mask(int32_t P, int32_t mode, const float *opacity, const float *vertex, const float *max_radii2D, float a, float b,
                    uint8_t *mask, void *stream);
/* _clipping_update_states (:330-345) on the masked rows: mode 0 opacity <- value; mode 1 vertex rescaled about its centre to mean side
 * `value` (:429-463); the rows' Adam moments <- 0 (exp_avg / exp_avg_sq may be NULL before the first optimizer step) */
int tsm_clip(int32_t P, int32_t mode, const uint8_t *mask, float value, float *param, float *exp_avg, float *exp_avg_sq, void *stream);
/* _opacity_reset (:524-537): opacity <- inverse_sigmoid(min(sigmoid(opacity), reset_value)); every row's Adam moments <- 0 */
int tsm_opacity_reset(int32_t P, float reset_value, float *opacity, float *exp_avg, float *exp_avg_sq, void *stream);
/* bg_depth of VanillaTSModel.forward (:623): out[0] <- max over the n_vertices rows of `vertex` (n_vertices, 3) of |camera_center - vertex|
 * (camera_center: 3 floats in DEVICE memory; out: 1 float in device memory; 0 for n_vertices == 0).  One read of the vertices instead of
 * torch's subtract / norm / max. */
int tsm_max_vertex_distance(int32_t n_vertices, const float *vertex, const float *camera_center, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
