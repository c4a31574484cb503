/*
 * lidargs_loss.h -- C ABI of the fused per-frame image loss.
 *
 * Replaces the ~45 framework ops of /root/reference/train.py:150-203 (+ utils/loss_utils.py:20-64): masked L1 on intensity
 * and depth, 10 x MSE on the ray-drop channel, 1 - SSIM (11 x 11 Gaussian window, sigma 1.5, zero padding) on the masked
 * intensity, and the masked L1 on horizontal depth differences -- and, because this is the root of the training graph, the
 * gradient of that scalar with respect to the rendered image and depth in the same call.
 *
 *   loss = depth_loss + (1 - lambda_dssim) Ll1 + lambda_dssim (1 - ssim) + raydrop_loss + grad_loss
 * (the reference adds scaling_reg = 0.01 mean(prod(scaling)), a per-Gaussian term that is not an image operation:
 *  lidargs_scaling_reg below, which can add its value onto losses[0]).
 *
 * image f32[2*H*W] (intensity, ray-drop), depth f32[H*W], gt f32[3*H*W] (ray-drop mask, intensity, depth): device pointers.
 * losses f32[6] (device): loss, Ll1, depth_loss, ssim_loss, raydrop_loss, grad_loss.
 * dL_dimage f32[2*H*W], dL_ddepth f32[H*W]: every element written.  Returns 0 or a negative LIDARGS_ERR_* code.
 */
#ifndef LIDARGS_LOSS_H
#define LIDARGS_LOSS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

size_t lidargs_loss_scratch_bytes(int H, int W);

int lidargs_image_loss(int H, int W, const float* image, const float* depth, const float* gt, float lambda_dssim,
                       float* losses, float* dL_dimage, float* dL_ddepth, char* scratch, size_t scratch_bytes, void* stream);

/*
 * scaling_reg = weight * mean(prod(scaling, dim=1)) of /root/reference/train.py:174 (weight 0.01 there), the per-Gaussian term of the
 * frame loss, and its gradient, in two launches with no host synchronisation (the framework's prod backward looks for zeros on the
 * host).  scaling f32[M*3] (the decode's `scaling` output), device.  value (nullable): the term; add_to (nullable): *add_to += term
 * (pass losses of lidargs_image_loss: losses[0] then is the reference's `loss`); dL_dscaling (nullable) f32[M*3]: every row
 * written, d(term)/d(scaling).  M = 0 gives NaN, as the mean of an empty tensor does.  The block sums are folded in a fixed order.
 */
size_t lidargs_scaling_reg_scratch_bytes(int M);

int lidargs_scaling_reg(int M, const float* scaling, float weight, float* value, float* add_to, float* dL_dscaling, char* scratch,
                        size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
