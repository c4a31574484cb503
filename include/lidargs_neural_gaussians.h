/*
 * lidargs_neural_gaussians.h -- C ABI of the fused anchor decode.
 *
 * Replaces the PyTorch graph of generate_neural_gaussians (/root/reference/gaussian_renderer/__init__.py:17-119): from the
 * anchors of a Scaffold-style model (feature, position, k offsets, 6 scalings) and four small MLPs
 * (scene/gaussian_model.py:113-142: Linear(35|36, 32)-ReLU-Linear(32, k | 7k | k | k) with tanh / identity / sigmoid /
 * sigmoid) to the packed per-Gaussian inputs of the rasterizer: xyz, colour (intensity, ray-drop), opacity, scaling, rotation,
 * keeping only the offsets whose neural opacity is > 0, in (anchor, offset) order.
 *
 * Supported model configuration (the reference's defaults, arguments/__init__.py:51-79): feat_dim 32, hidden 32,
 * n_offsets in {4,5,6,8,10}, appearance_dim 0, use_feat_bank false, colour channels 2.  Anything else is the caller's job to
 * reject (the Python front-end does, loudly).
 *
 * All pointers are device pointers unless stated otherwise; plain float32 row-major arrays; no torch types.  Functions return 0
 * or a positive count on success, a negative LIDARGS_ERR_* code on failure (message: lidargs_last_error()).
 */
#ifndef LIDARGS_NEURAL_GAUSSIANS_H
#define LIDARGS_NEURAL_GAUSSIANS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The four MLPs, in the order opacity, cov, color, raydrop.  Weights in nn.Linear layout: W1 [32][din], b1 [32],
 * W2 [dout][32], b2 [dout] with din = 35 + add_*_dist and dout = k, 7k, k, k.  W2T [32][dout] is W2 transposed: required by
 * the backward entry points; the forward ones use it when given (their MLPs then run as tile products on the matrix pipe, with
 * bit-identical results) and fall back to one anchor per lane when it is NULL. */
typedef struct lidargs_ng_model {
    int n_offsets;
    int add_opacity_dist, add_cov_dist, add_color_dist;
    const float* W1[4]; const float* b1[4]; const float* W2[4]; const float* b2[4];
    const float* W2T[4];
} lidargs_ng_model;

size_t lidargs_ng_scratch_bytes(int N, int n_offsets);

/* Step 1 (:19-68): visible-anchor selection, view direction / distance, opacity MLP, opacity > 0 mask.
 *   visible_mask   u8[N] (torch.bool) or NULL = all visible
 *   cam_center     HOST pointer, 3 floats (viewpoint_camera.camera_center)
 *   neural_opacity f32[N*k]  written for the first n*k entries: tanh output per (visible anchor, offset), anchor order
 *   mask           u8[N*k]   first n*k entries: neural_opacity > 0
 *   counts_host    HOST int[2]: n = number of visible anchors, M = number of selected (anchor, offset) pairs
 * Synchronises the stream once (the counts size the outputs of step 2, like the reference's boolean indexing does). */
int lidargs_ng_forward_select(int N, const lidargs_ng_model* model, const uint8_t* visible_mask, const float* anchor_feat,
                              const float* anchor, const float* cam_center, float* neural_opacity, uint8_t* mask,
                              int* counts_host, char* scratch, size_t scratch_bytes, void* stream);
/* The same without the wait: the two counts are copied to `counts_pinned` (PINNED host memory, int[2]) by the stream and the call
 * returns at once.  Step 2 does not need them -- its output rows come from the device-side scan -- so a caller that gives step 2
 * arrays of N*k rows each (the upper bound of M) can queue it right behind this call and wait for the counts (an event recorded
 * after this call) while it runs, then take the first M rows: the device never idles between the two steps. */
int lidargs_ng_forward_select_enqueue(int N, const lidargs_ng_model* model, const uint8_t* visible_mask, const float* anchor_feat,
                                      const float* anchor, const float* cam_center, float* neural_opacity, uint8_t* mask,
                                      int* counts_pinned, char* scratch, size_t scratch_bytes, void* stream);

/* Step 2 (:70-113): colour / ray-drop / covariance MLPs and the post-processing, for the selected pairs only.
 * Outputs have M rows: xyz f32[M*3], color f32[M*2], opacity f32[M], scaling f32[M*3], rot f32[M*4].
 * `scaling` input is get_scaling (already exp-activated) f32[N*6]; offset f32[N*k*3]; neural_opacity is what step 1 wrote.
 * Must follow lidargs_ng_forward_select of the same inputs with the same scratch (the selection lives there). */
int lidargs_ng_forward_decode(int N, const lidargs_ng_model* model, const float* anchor_feat, const float* anchor,
                              const float* offset, const float* scaling, const float* cam_center, const float* neural_opacity,
                              float* out_xyz, float* out_color, float* out_opacity, float* out_scaling, float* out_rot,
                              char* scratch, size_t scratch_bytes, void* stream);

/* Backward.  Upstream gradients of the M-row outputs; dL_dneural_opacity f32[n*k] (gradient of the un-masked neural_opacity
 * output) may be NULL -- the reference only uses that output as a statistic (train.py:243).  Writes, for every one of the N anchors (zeros for
 * invisible ones): dL_danchor_feat f32[N*32], dL_danchor f32[N*3], dL_doffset f32[N*k*3], dL_dscaling f32[N*6]; and for
 * the n visible anchors, in anchor order, the per-anchor layer inputs and deltas the weight gradients are plain GEMMs of:
 *   act_x  f32[n][40]   the 36 inputs (feature, view, dist), a constant 1, three zeros
 *   act_h  f32[n][132]  the ReLU outputs of the four MLPs (opacity, cov, color, raydrop: 32 each), a constant 1, three zeros
 *   delta1 f32[n][128]  hidden-layer deltas of the four MLPs;   delta2 f32[n][10k]  output deltas laid out [k | 7k | k | k]
 * so that  G1 = delta1^T act_x  [128 x 40]  holds dW1_m in rows 32m..32m+31, columns 0..din_m-1, and db1_m in column 36;
 *          G2 = delta2^T act_h  [10k x 132] holds dW2_m in its row block, columns 32m..32m+31, and db2_m in column 128.
 * Must follow forward_select/forward_decode of the same inputs with the same scratch (the selection is reused). */
int lidargs_ng_backward(int N, int n_visible, const lidargs_ng_model* model, const float* anchor_feat, const float* anchor,
                        const float* offset, const float* scaling, const float* cam_center,
                        const float* dL_dxyz, const float* dL_dcolor, const float* dL_dopacity, const float* dL_dscaling,
                        const float* dL_drot, const float* dL_dneural_opacity, float* dL_danchor_feat, float* dL_danchor, float* dL_doffset,
                        float* dL_dscaling_in, float* act_x, float* act_h, float* delta1, float* delta2,
                        char* scratch, size_t scratch_bytes, void* stream);

/* Backward on the matrix pipe (f32 in, f32 accumulate: v_mfma_f32_16x16x4_f32 for k <= 6, v_mfma_f32_32x32x2_f32 for k = 8, 10): recompute, back-propagation and the weight gradients of the four
 * MLPs as 32-row tile products; nothing per-anchor is written for a GEMM to read.  Same inputs and dense outputs as
 * lidargs_ng_backward; instead of act_x / act_h / delta1 / delta2 the kernel's persistent waves write partial sums:
 * partials f32[waves][floats_per_wave] (sizes from lidargs_ng_backward_partials; `waves` = the number of partial ROWS: round 4 one per
 * wave, 4 x CUs; round 5 (k <= 6: 16x16x4 tiles, two launches) one per workgroup and launch, 2 x CUs -- a caller sizes the array and
 * calls lidargs_ng_reduce_weight_grads with whatever this function reports), and the SUM OVER WAVES is
 *   tiles  f32[5 + T2][32][32], T2 = 3 + ceil(7k/32):
 *     tile m (m = 0 opacity, 1 covariance, 2 colour, 3 ray-drop)  [t][q]  dW1_m[t][q], q = 0..31
 *     tile 4   [t][8 m + j]   dW1_m[t][32 + j], j = 0..3, and db1_m[t] at j = 4 (input 36 is the constant 1)
 *     tile 5   [o][t] dW2 of the opacity MLP (rows o < k);  tiles 6 .. 5 + ceil(7k/32): covariance, row o of tile o / 32;
 *     then colour, then ray-drop
 *   db2    f32[128] after the tiles: output-layer bias gradients laid out [k | 7k | k | k]. */
int lidargs_ng_backward_partials(int n_offsets, int* waves, int* floats_per_wave);
int lidargs_ng_backward_mfma(int N, const lidargs_ng_model* model, const float* anchor_feat, const float* anchor,
                             const float* offset, const float* scaling, const float* cam_center,
                             const float* dL_dxyz, const float* dL_dcolor, const float* dL_dopacity, const float* dL_dscaling,
                             const float* dL_drot, const float* dL_dneural_opacity, float* dL_danchor_feat, float* dL_danchor, float* dL_doffset,
                             float* dL_dscaling_in, float* partials, char* scratch, size_t scratch_bytes, void* stream);

/* model.W2T of all four MLPs in one launch: W2[m] = the second Linear's weight [dout_m][32] (device pointers, host array of four);
 * out f32[320 k] = [32][k] | [32][7k] | [32][k] | [32][k], i.e. model.W2T[m] = out + 32 * (0, k, 8k, 9k)[m]. */
int lidargs_ng_transpose_w2(int n_offsets, const float* const* W2, float* out, void* stream);

/* The sixteen parameter gradients out of lidargs_ng_backward_mfma's partials, in two small launches (sum over the waves + unpack; what the
 * binding did with a framework reduction, four concatenations and four strided copies).  din[m] = input width of MLP m's first layer
 * (35 or 36: 32 features + 3 view components [+ 1 distance]); waves as lidargs_ng_backward_partials reported.
 * grads f32[lidargs_ng_weight_grad_floats(k, din)], for m = opacity, covariance, colour, ray-drop in turn:
 *   dW1_m [32][din_m] | db1_m [32] | dW2_m [dout_m][32] | db2_m [dout_m],   dout = k, 7k, k, k
 * -- the layouts of torch.nn.Linear's weight and bias (gaussian_model.py:113-142), so each block is that parameter's .grad.
 * stage: lidargs_ng_weight_grad_stage_floats(k, din) floats of scratch (the first launch's row-group sums).  Sums are taken in a fixed order (deterministic). */
int lidargs_ng_weight_grad_floats(int n_offsets, const int* din);
int lidargs_ng_weight_grad_stage_floats(int n_offsets, const int* din);
int lidargs_ng_reduce_weight_grads(int n_offsets, const int* din, int waves, const float* partials, float* grads, float* stage, void* stream);

/* Densification statistics -- GaussianModel.training_statis (scene/gaussian_model.py:599-622), in place, one launch chain, no host
 * read.  anchor_visible_mask u8[N]; offset_selection_mask u8[n*k] and neural_opacity f32[n*k] in visible-anchor order (what
 * generate_neural_gaussians returned); update_filter u8[M] (radii > 0) and viewspace_grad f32[M*4] (means2D.grad) in the order of
 * the selected pairs.  Updates opacity_accum f32[N] += sum_j max(opacity, 0), anchor_demon f32[N] += 1 (visible anchors),
 * offset_gradient_accum f32[N*k] += |grad[2:4]| and offset_denom f32[N*k] += 1 (selected pairs whose Gaussian was on screen).
 * scratch: lidargs_ng_scratch_bytes(N, k). */
int lidargs_ng_training_stats(int N, int n_offsets, const uint8_t* anchor_visible_mask, const uint8_t* offset_selection_mask,
                              const uint8_t* update_filter, const float* neural_opacity, const float* viewspace_grad,
                              float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                              char* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
