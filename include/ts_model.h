/* ts_model.h -- C ABI of the per-iteration model-update statistics, exported by libts2d.so.
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

#ifdef __cplusplus
}
#endif
#endif
