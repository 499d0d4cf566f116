/* ts_loss.h -- C ABI of the fused photometric loss (L1 + SSIM) that feeds the rasterizer's dL_dout_feature.
 *
 * the step either side of the hot path.  Replaces, for one image pair, the reference's
 *     l1_loss   = L1(image, gt_image)                       src/diff_recon/trainers/trainer_utils.py:323-324
 *     ssim_loss = ssimLoss(image, gt_image)                 trainer_utils.py:9-103 (GaussianSmoothing2D 11x11 sigma 1.5,
 *                                                           zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements)
 *     img_loss  = w_L1 * l1_loss + w_ssim * ssim_loss       src/diff_recon/trainers/VanillaTS_trainer.py:80-81,111
 * and their autograd backward (ten eager torch kernels forward, more backward) with two HIP kernels each way.
 * Exported by the same libts2d.so as include/ts2d.h.  All pointers are device pointers; images are planar float32
 * (C, H, W) -- a leading batch dimension folds into C, exactly as the reference's depthwise conv + global mean does.
 * Every call enqueues on `stream` and returns TS2D_OK (0) or an error code of ts2d.h with ts2d_last_error() set. */
#ifndef TS_LOSS_H
#define TS_LOSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device scratch the forward fills and the backward reads (three per-element derivative maps + per-block
 * partial sums). */
size_t tsl_workspace_bytes(int32_t channels, int32_t height, int32_t width);

/* Forward.  out[0] = w_l1 * L1 + w_ssim * (1 - SSIM), out[1] = L1 = mean|image - gt|, out[2] = 1 - mean(ssim_map)
 * (three floats in device memory).  `need_grad` = 0 skips writing the derivative maps (evaluation). */
int tsl_photometric_forward(const float *image, const float *gt, int32_t channels, int32_t height, int32_t width,
                            float w_l1, float w_ssim, int32_t need_grad, void *workspace, size_t workspace_bytes, float *out,
                            void *stream);

/* Backward of out[0] with respect to `image`: dL_dimage (C, H, W) fully written.  `grad_out` is a device scalar
 * (upstream gradient of the loss) or NULL for 1.  `workspace` must be the buffer the matching forward filled. */
int tsl_photometric_backward(const float *image, const float *gt, int32_t channels, int32_t height, int32_t width,
                             float w_l1, float w_ssim, const void *workspace, size_t workspace_bytes, const float *grad_out,
                             float *dL_dimage, void *stream);

/* ---- depth / normal consistency loss ------------------------------------------------------------------------------------
 * Replaces DepthNormalLoss (src/diff_recon/trainers/trainer_utils.py:204-257; geometry_loss of VanillaTS_trainer.py:30-31,84,111 --
 * w_geometry = 0.05, scale_factor = 0.5 in the *_VanillaTS_mesh.yaml configurations): the producer of the rasterizer's
 * dL_dout_depth / dL_dout_normal.  depth (H, W) and normal (3, H, W) are the rasterizer's rich_info outputs (planar float32,
 * device memory).
 *     d = bilinear(depth, scale_factor);  (gx, gy) = Scharr(d);  raw normal from (gx, gy) / d and the pinhole (tan_fovx, tan_fovy);
 *     Dn = normalize(bilinear(raw normal, (H, W)));  mask = bilinear(|(gx, gy)|) < quantile(.., depth_grad_filter_quantile);
 *     loss = mean((1 - <normalize(normal, eps 1e-8), Dn>) * mask)
 * scale_factor <= 0 or == 1 means "no resampling" (the reference's scale_factor = None); it is a double, like the Python float
 * F.interpolate receives: the low-resolution size floor(H * s) and the coordinate ratio (float)(1 / s) are formed from it in double. */
size_t tsl_depth_normal_workspace_bytes(int32_t height, int32_t width, double scale_factor);

/* Forward: out[0] = loss (one float in device memory).  The workspace keeps what the backward needs. */
int tsl_depth_normal_forward(const float *depth, const float *normal, int32_t height, int32_t width, float tan_fovx, float tan_fovy,
                             double scale_factor, float quantile, void *workspace, size_t workspace_bytes, float *out, void *stream);

/* Backward of out[0]: dL_ddepth (H, W) and / or dL_dnormal (3, H, W), fully written; either may be NULL (the reference's depth_grad /
 * normal_grad = False).  `grad_out`: device scalar or NULL for 1.  `workspace`: the buffer the matching forward filled. */
int tsl_depth_normal_backward(const float *depth, const float *normal, int32_t height, int32_t width, float tan_fovx, float tan_fovy,
                              double scale_factor, const void *workspace, size_t workspace_bytes, const float *grad_out, float *dL_ddepth,
                              float *dL_dnormal, void *stream);

/* ---- the two auxiliary image losses -------------------------------------------------------------------------------------------
 * DoGLoss(freq, scale_factor) and SmoothnessLoss(quantile, scale_factor), src/diff_recon/trainers/trainer_utils.py:105-148, 181-201 (w_dog,
 * w_smoothness of VanillaTS_trainer.py:26-29, 82-83, 111; 0 in every shipped configuration).  Images are planar float32 (C, H, W), C <= 8;
 * a mask is ONE plane (H, W) of 0 / 1 floats formed from the TARGET image without gradient:
 *   tsl_dog_mask:        grey = mean_c(gt); d = bilinear(grey, scale_factor); DoG = gauss(d, sigma1, ksize1) - gauss(d, sigma2, ksize2)
 *                        (zero padding); U = bilinear(DoG, (H, W)); n = (U - min U) / (max U - min U), 1 - n when `invert` (freq >= 50);
 *                        mask = n >= 0.5.  The caller passes the kernel sizes (Python: int(2 * round(3 * sigma) + 1)).
 *   tsl_smoothness_mask: d = bilinear(gt, scale_factor) per channel; g = |Scharr(d)|_2 over the 2 C gradient planes; U = bilinear(g, (H, W));
 *                        mask = U < quantile(U, q) (torch.quantile, linear).
 * scale_factor as in tsl_depth_normal_* (a double; <= 0 or 1 = no resampling).  One workspace size serves all six calls. */
size_t tsl_aux_loss_workspace_bytes(int32_t channels, int32_t height, int32_t width, double scale_factor);
int tsl_dog_mask(const float *gt, int32_t channels, int32_t height, int32_t width, double sigma1, int32_t ksize1, double sigma2, int32_t ksize2,
                 int32_t invert, double scale_factor, void *workspace, size_t workspace_bytes, float *mask, void *stream);
int tsl_smoothness_mask(const float *gt, int32_t channels, int32_t height, int32_t width, double scale_factor, float quantile, void *workspace,
                        size_t workspace_bytes, float *mask, void *stream);
/* out[0] = L1(image * mask, gt * mask) = mean over C H W of |image m - gt m| (trainer_utils.py:147-148); backward: dL_dimage fully written
 * (sign(0) = 0 like torch's abs), `grad_out` a device scalar or NULL for 1. */
int tsl_masked_l1_forward(const float *image, const float *gt, const float *mask, int32_t channels, int32_t height, int32_t width, void *workspace,
                          size_t workspace_bytes, float *out, void *stream);
int tsl_masked_l1_backward(const float *image, const float *gt, const float *mask, int32_t channels, int32_t height, int32_t width,
                           const float *grad_out, float *dL_dimage, void *stream);
/* out[0] = mean over H W of |Scharr(image)|_2 * mask (trainer_utils.py:196-200; zero-padded Scharr pair per channel, norm over the 2 C planes,
 * gradient 0 where the norm is 0 like torch's norm); backward: dL_dimage (C, H, W) fully written. */
int tsl_scharr_smoothness_forward(const float *image, const float *mask, int32_t channels, int32_t height, int32_t width, void *workspace,
                                  size_t workspace_bytes, float *out, void *stream);
int tsl_scharr_smoothness_backward(const float *image, const float *mask, int32_t channels, int32_t height, int32_t width, void *workspace,
                                   size_t workspace_bytes, const float *grad_out, float *dL_dimage, void *stream);

/* ---- the down-sampler of render_up_scale ----------------------------------------------------------------------------------------------
 * F.interpolate(x, size=(h, w), mode="bilinear") of VanillaTS_model.py:649-656 for an INTEGER factor (H = f h, W = g w, f, g >= 2 -- render_up_scale
 * = 2 in the NerfSynthetic *_mesh configuration): planar float32 (C, H, W) -> (C, h, w), PyTorch's source-index arithmetic in float32, and its
 * backward in gather form (torch's upsample_bilinear2d_backward scatters with atomics): both fully written, run-to-run identical. */
int tsl_downsample_forward(const float *in, int32_t channels, int32_t in_height, int32_t in_width, int32_t out_height, int32_t out_width, float *out,
                           void *stream);
int tsl_downsample_backward(const float *grad_out, int32_t channels, int32_t in_height, int32_t in_width, int32_t out_height, int32_t out_width,
                            float *grad_in, void *stream);
/* The same resize for planes that are NOT one tensor -- the render (3), depth (1) and normal (3) images of VanillaTS_model.py:649-656 are three:
 * `in_planes` / `out_planes` (backward: `grad_out_planes` / `grad_in_planes`) are HOST arrays of num_planes device pointers, one per image plane;
 * up to TS_RESAMPLE_PLANES planes share a launch (at 800 x 800 a launch is latency, not work: six launches per iteration become two). */
#define TS_RESAMPLE_PLANES 8
int tsl_downsample_forward_planes(int32_t num_planes, const float *const *in_planes, int32_t in_height, int32_t in_width, int32_t out_height,
                                  int32_t out_width, float *const *out_planes, void *stream);
int tsl_downsample_backward_planes(int32_t num_planes, const float *const *grad_out_planes, int32_t in_height, int32_t in_width, int32_t out_height,
                                   int32_t out_width, float *const *grad_in_planes, void *stream);

#ifdef __cplusplus
}
#endif
#endif
