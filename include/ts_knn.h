/* ts_knn.h -- C ABI of the exact nearest-neighbour helpers, exported by libts2d.so.
 *
 * Replaces submodules/simple-knn (the reference's `simple_knn._C`):
 *     distCUDA2(points)                    interface.cu:6-26   -> SimpleKNN::knn             simple_knn.cu:237-283
 *     nearestNeighbor(points, batch_size)  interface.cu:28-52  -> SimpleKNN::nearestNeighbor simple_knn.cu:285-330
 * callers: src/diff_recon/models/model_utils.py:36 (initial triangle size), src/diff_recon/trainers/VanillaTS_trainer.py:108
 * (geometry loss, batch_size 3 = the three vertices of a triangle).  Both are EXACT searches, so results are defined by
 * the mathematics, not by the traversal; the traversal here is MI355X-shaped (Morton-sorted points physically gathered,
 * one workgroup per 1024-point box, candidate boxes staged through LDS and pruned per workgroup and per lane).
 * All pointers are device pointers; calls enqueue on `stream` and never synchronise. */
#ifndef TS_KNN_H
#define TS_KNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device scratch for P points. */
size_t tsk_workspace_bytes(int32_t P);

/* mean_dist2[i] = mean of the squared distances from points[i] to its 3 nearest OTHER points (simple_knn.cu:153-187).
 * With fewer than 4 points the missing neighbours count as FLT_MAX like the reference (the sum overflows to +inf for
 * P <= 2).  points: P*3 floats, mean_dist2: P floats. */
int tsk_mean_dist3(int32_t P, const float *points, float *mean_dist2, void *workspace, size_t workspace_bytes, void *stream);

/* nearest[i] = index of the nearest point j with j / batch_size != i / batch_size (simple_knn.cu:189-235).  Ties go to
 * the candidate that comes first in (Morton code, index) order, which is what the reference's strict `<` over its
 * Morton-ordered scan yields.  If no candidate exists (P == batch_size) nearest[i] = i (the reference leaves it
 * undefined).  batch_size > 0 and P % batch_size == 0 are the caller's checks (interface.cu:30-37). */
int tsk_nearest_other(int32_t P, int32_t batch_size, const float *points, uint32_t *nearest, void *workspace,
                      size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
