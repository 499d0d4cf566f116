/* ts_optim.h -- C ABI of the optimizer step of the training loop, exported by libts2d.so ("reduce-scatter + sharded Adam +
 * all-gather"; VERDICT r3: the next-largest cost of a step after the rasterizer once the loss is fused).
 *
 * Replaces `self.model.optimizer.step()` of the reference's trainer (src/diff_recon/trainers/VanillaTS_trainer.py:119-122), where the
 * optimizer is torch.optim.Adam over four per-triangle parameter groups -- vertex (P,3,3), opacity (P,1), f_dc (P,1,3), f_rest (P,M-1,3) --
 * with per-group learning rates set every iteration, betas (0.9, 0.999), eps = 1e-15, no weight decay, no amsgrad
 * (src/diff_recon/models/VanillaTS_model.py:108-124, update_learning_rate :583).  torch runs that as ~10 multi-tensor launches per step;
 * here it is ONE launch over any number of flat ranges ("slices"): all four groups of a single-GPU step, or the parts of the groups that
 * fall into the 1 / world slice of a flat parameter buffer that a rank owns after the gradients' reduce-scatter (sharded Adam).
 *
 * Arithmetic per element, fp32, in torch's operation order (torch/optim/adam.py, _single_tensor_adam):
 *     g            = grad * grad_scale                      (grad_scale = 1, or 1 / world for mean-reduced gradients)
 *     exp_avg      = exp_avg + (g - exp_avg) * (1 - beta1)                                  (lerp_; for beta1 <= 0.5 ATen's other branch:
 *                    g - (g - exp_avg) * (1 - (1 - beta1)))
 *     exp_avg_sq   = exp_avg_sq * beta2 + (1 - beta2) * g * g                               (mul_, addcmul_)
 *     denom        = sqrt(exp_avg_sq) / bias2_sqrt + eps        bias2_sqrt = sqrt(1 - beta2^t)
 *     param        = param - step_size * (exp_avg / denom)      step_size  = lr / (1 - beta1^t)
 * The two bias corrections are formed by the caller in double, like torch forms them in Python, and handed over as floats.  Built without
 * FMA contraction; torch's own kernels may contract a product into the following sum, so results agree with torch.optim.Adam to 1 ulp per
 * operation, not bit for bit (tests/test_optim_gpu.py states the bound).  HBM-bound: 16 bytes read + 12 written per element.
 * All pointers are device pointers; `count` floats each; asynchronous on `stream`. */
#ifndef TS_OPTIM_H
#define TS_OPTIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tso_adam_slice
{
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t count;     /* floats in this slice */
    float step_size;   /* lr / (1 - beta1^t) */
    float bias2_sqrt;  /* sqrt(1 - beta2^t) */
    float grad_scale;  /* multiplies the gradient (1 = as is) */
    /* optional second learning rate inside ONE tensor: elements whose index (index0 + i) % period >= split use step_size_tail.  period = 0
     * switches it off.  This is how a single (P, M, 3) SH tensor carries the reference's two groups f_dc (first 3 floats of every triangle's
     * 3 M) and f_rest (the other 3 M - 3) without the torch.cat of VanillaTS_model.py:79-80 in every forward. */
    float step_size_tail;
    int64_t index0;    /* index of this slice's first element inside its tensor (a rank's slice starts in the middle of one) */
    int32_t period, split;
} tso_adam_slice;

#define TSO_MAX_SLICES 16 /* per call */

/* beta1, beta2, eps are doubles like the Python floats torch receives: 1 - beta is formed in double and THEN rounded to fp32 (1 - 0.999f is
 * 1.3e-5 off 0.001). */
int tso_adam_step(const tso_adam_slice *slices, int32_t num_slices, double beta1, double beta2, double eps, void *stream);

/* The Adam step of the SH coefficients from their FACTORED gradient.  In SH mode every view's dL/dshs is a rank-1 product per triangle,
 *     dL_dshs[i, k, :] = sum_v basis_k(normalize(centre_i - campos_v)) * dL_dRGB_v[i, :]          (R2D/src/backward.cu:9-119),
 * and ts2d_backward under TS2D_FLAG_SH_FACTORED leaves the (P, M, 3) array unwritten and returns dL_dRGB (P, 3) in its dL_dfeature slot.  This
 * call forms the product in registers and applies the arithmetic above to it: bit-identical parameters and moments to tso_adam_step on the dense
 * gradient, with 12 M bytes per triangle less written by the backward and 12 M - 12 V less read here.
 * `vertex` must hold the values the backward passes ran on: call this BEFORE the step that updates the vertices.
 * Coefficient 0 of triangle i lives at {param,exp_avg,exp_avg_sq}_dc + i * dc_stride, coefficient k >= 1 at ..._rest + i * rest_stride + 3 (k - 1)
 * (strides in floats): the reference's two tensors f_dc (P, 1, 3) / f_rest (P, M - 1, 3) are strides 3 / 3 (M - 1); ONE (P, M, 3) tensor is
 * dc = base, rest = base + 3, both strides 3 M.  Coefficients above sh_degree take a zero gradient (the reference's dense array holds zeros there). */
typedef struct tso_row_slice
{
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int32_t floats_per_row;   /* 9 for (P, 3, 3) vertices, 1 for (P, 1) opacities */
    float step_size;
    float bias2_sqrt;
    float grad_scale;
} tso_row_slice;
#define TSO_SH_ROW_SLICES 2

typedef struct tso_sh_factored_step
{
    int32_t P, M, sh_degree, V;   /* M = (max degree + 1)^2 in {1, 4, 9, 16}; V views >= 1 */
    const float *vertex;          /* P*9 */
    const float *campos;          /* V*3, device memory */
    const float *dL_dcolor;       /* V*P*3 */
    float *param_dc, *exp_avg_dc, *exp_avg_sq_dc;
    float *param_rest, *exp_avg_rest, *exp_avg_sq_rest; /* may be NULL for M == 1 */
    int64_t dc_stride, rest_stride;
    float step_size_dc, bias2_sqrt_dc, step_size_rest, bias2_sqrt_rest;
    float grad_scale;
    /* optional: further per-triangle parameters (the vertices, the opacities) stepped from their DENSE gradients by the same launch -- the workgroup
     * that owns 64 triangles updates their rows after it has read what it needs of them (the direction comes from the vertices BEFORE the step):
     * one launch for the whole optimizer step, and the vertices are read once.  floats_per_row * P floats each. */
    int32_t num_rows;             /* 0 .. TSO_SH_ROW_SLICES */
    tso_row_slice rows[TSO_SH_ROW_SLICES];
} tso_sh_factored_step;

int tso_adam_step_sh_factored(const tso_sh_factored_step *step, double beta1, double beta2, double eps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TS_OPTIM_H */
