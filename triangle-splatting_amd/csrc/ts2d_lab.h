/*
 * ts2d_lab.h -- entry points that exist ONLY in tools/bin/libts2d_lab.so (python triangle-splatting_amd/build.py --lab): diagnostics and
 * comparators for tests/ and tools/.  The product library libts2d.so exports nothing of this (tests/test_cabi_cpu.py checks its export
 * table against include/*.h) and links no rocPRIM.  The lab library contains the product's objects, so it reads the private state of a
 * forward that the PRODUCT library ran in the same process (same layout code, device pointers are process-wide).
 */
#ifndef TS2D_LAB_H
#define TS2D_LAB_H

#include "../../include/ts2d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Copies field `field` of the private state into host memory `dst` (dst_bytes must be large enough), synchronising `stream`.  Fields:
 *   0 screen verts (P*6 f32: v1.xy v2.xy v3.xy)   1 area2 (P f32)       2 normal_view (P*3 f32)
 *   3 v_depth (P*3 f32)   4 depth key (P f32)      5 rgb (P*3 f32)       6 clamped (P u8, bits 0..2)
 *   7 instance offsets in depth order (P u32)  8 tiles_touched (P u32)  9 rect (P*4 u32: minx miny maxx maxy)
 *   10 sorted keys (N u64)   11 sorted triangle ids (N u32)   12 ranges (T*2 u32)
 *   13 n_contrib (H*W u32)   14 final_T (H*W f32)   15 / 16 the ping-pong partner of the sorted instance list (N u32 each)
 *   17 triangle ids in (depth, id) order (P u32)   18 raw render records (P*16 f32; with TS2D_FLAG_3D: v1_view v2_view
 *   v3_view normal_view opacity rgb -- fields 0-3 and 5 decode the 2D record layout only) */
int ts2d_debug_read_state(const ts2d_state *state, int32_t P, int64_t num_rendered, int32_t width, int32_t height,
                          int32_t field, void *dst, size_t dst_bytes, void *stream);

/* The binning primitives that replace cub::DeviceRadixSort::SortPairs / cub::DeviceScan::InclusiveSum (R2D/src/rasterizer.cu:210-218, 186)
 * on caller-provided device arrays: the hand-written stable LSD radix sort of (key, value) pairs on bits [0, end_bit) (which = 0;
 * csrc/binning.hip -- which = 2: with the hierarchical passes that sorts of more than 48 slabs take) and AMD's rocPRIM on the same arrays
 * (which = 1; the comparator).  n pairs, synchronous. */
int ts2d_test_sort_pairs(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out, uint32_t *vals_out, size_t n,
                         int32_t end_bit, int32_t which, void *stream);
int ts2d_test_inclusive_scan_rocprim(const uint32_t *in, uint32_t *out, size_t n, void *stream);

/* on != 0: every sort of later forwards IN THIS LIBRARY takes the hierarchical (ticket) passes that sorts of more than 48 slabs take (> 6.3 M
 * triangles, > 12.6 M instances) and the scan its two-level form (> 2 M triangles), whatever the scene's size -- so that the suite executes
 * them (ticket-path depth census included). */
void ts2d_lab_force_ticket_passes(int on);
/* on != 0: the depth sort of later forwards in this library always runs its fourth pass.  The product skips it when all visible depths
 * share the top key byte (sign + 7 exponent bits: depths within a factor of four -- every synthetic scene of bench.py; not a real scene
 * that spans more): bench.py --force-depth-pass4 reports the headline without that data-dependent shortcut. */
void ts2d_lab_force_depth_pass4(int on);
size_t ts2d_test_quantile_scratch_bytes(void);
int ts2d_test_quantile(const uint32_t *keys, size_t n, float q, void *scratch, float *out, void *stream); // torch.quantile by the library's radix select
void ts2d_lab_depth_split(int mode, int bucket_cap); // mode 0: the product's choice (sampled splitters from 12 289 to 500 000 triangles), 1: the LSD depth sort at every size, 2: sampled splitters up to 1.6 M; bucket_cap > 0: a smaller per-bucket register capacity (the kernel's global-memory path)
/* on != 0: the emission kernel of later forwards in this library flags EVERY quadrant of every instance (the masks' machinery runs, the test
 * always passes).  The masks are pure culling of work that contributes nothing, so every output must be what it is with them on:
 * tests/test_qmask_gpu.py compares the two, bit for bit where the arithmetic is ordered. */
void ts2d_lab_force_all_quadrants(int on);

/* Round 6, a measured negative result kept reproducible (csrc/api.hip: SideLane; profiles/r06_side_stream.txt): on != 0 -> SH scenes of 131 072
 * triangles and more run the per-triangle kernel WITHOUT the SH colours and evaluate those with a throttled kernel on a library-owned stream,
 * forked behind the per-triangle kernel and joined in front of the blend kernel (bench.py --side-stream [--colour-blocks n]; n = the colour
 * kernel's grid = its throttle).  State and outputs are those of the single launch, bit for bit (tests/test_side_stream_gpu.py). */
void ts2d_lab_side_stream(int on);
void ts2d_lab_colour_blocks(int blocks);

#ifdef __cplusplus
}
#endif
#endif /* TS2D_LAB_H */
