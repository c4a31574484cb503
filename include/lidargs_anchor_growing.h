/*
 * lidargs_anchor_growing.h -- C ABI of native anchor growing (second half).
 *
 * Replaces the body of the loop of GaussianModel.anchor_growing (/root/reference/scene/gaussian_model.py:677-775, called from
 * adjust_anchor :776-784 every `update_interval` iterations, train.py:246-247).  Per level the reference
 *   (:683-688) builds the candidate mask  grads >= threshold * 2^i  &  offset_mask  &  rand > 0.5^(i+1)  over the N0*k offsets that
 *              existed when the call started,
 *   (:698-709) quantises every anchor and every candidate offset position  anchor + offset * scaling[:, :3]  to voxels of edge
 *              cur_size = voxel_size * 16 / 4^i  with  round(x / cur_size).int(),
 *   (:711)     torch.unique(dim=0, return_inverse=True) of the candidates' voxel rows,
 *   (:714-727) drops the voxels that already hold an anchor by comparing EVERY (candidate voxel, anchor) pair in chunks of 4096 anchors
 *              -- O(candidates x anchors): ~1e10 integer compares at 1e4 candidate voxels x 1.2 M anchors,
 *   (:730)     new anchor positions = voxel * cur_size, in the sorted order torch.unique left,
 *   (:740-742) torch_scatter.scatter_max of the candidates' 32-wide anchor features per voxel.
 * Here: one pass marks the candidates and folds their voxel bounding box; the voxels go into an open-addressing hash set (64-bit keys
 * packed to the bounding box, so any int32 voxel coordinates are exact); the existing anchors probe it (O(anchors), replacing the pair
 * scan); the surviving keys are radix-sorted (x most significant = torch.unique's row order) and the feature maximum is an integer
 * atomicMax on an order-preserving image of the float.  Everything the reference produces per level that is not a constant fill
 * (new_scaling, new_rotation, new_opacities, new_offsets are: :734-747) comes out of this one call.
 *
 * All array arguments are DEVICE pointers, float32 / uint8, contiguous.  Returns the number of new anchors U (>= 0) or a negative
 * LIDARGS_ERR_* code (message: lidargs_last_error()).  The call waits for the stream twice (the candidate count and U size the
 * buffers, like the reference's boolean indexing and its `if candidate_anchor.shape[0] > 0` do).
 */
#ifndef LIDARGS_ANCHOR_GROWING_H
#define LIDARGS_ANCHOR_GROWING_H

#include <stddef.h>
#include <stdint.h>

#include "lidargs_rasterizer.h" /* lidargs_alloc_fn, error codes */

#ifdef __cplusplus
extern "C" {
#endif

/* flags */
#define LIDARGS_AG_EXACT_DIVISION 1 /* quotient x / float32(cur_size) as an IEEE division (what torch's CPU kernel computes and the golden
                                       fixture pins); default 0: x * float32(1.0 / cur_size), the reciprocal of the Python DOUBLE rounded to
                                       float32 -- what torch's DEVICE kernel computes for `tensor / python_scalar` (ATen
                                       BinaryDivTrueKernel.cu), i.e. what the reference computes where it really runs.  Measured bit for bit
                                       against torch-ROCm on the MI355X for eight voxel sizes (tools/div_convention.py,
                                       profiles/r06_div_convention.txt): 4 M quotients each, 0 differences; the IEEE quotient puts 7-600 of
                                       them into another voxel, the float32 reciprocal 44-900 where 1/float32(s) != float32(1/s) */

/* Bytes of the fixed scratch buffer: the candidate list (one word per offset that existed at the start) + counters. */
size_t lidargs_ag_scratch_bytes(int N0, int n_offsets);

/* One level of anchor growing.
 *   N              anchors NOW (including those grown by earlier levels of the same call);  N0 <= N: anchors when the call started
 *   anchor         f32[N,3]      get_anchor (:255)
 *   offset         f32[N,k,3]    _offset
 *   scaling        f32[N,6]      get_scaling = exp(_scaling) (:213-214); columns 0..2 are used (:698)
 *   anchor_feat    f32[N,F]      _anchor_feat (F = feat_dim <= 256)
 *   grads          f32[N0*k]     norm of the accumulated offset gradient (:779-780)
 *   offset_mask    u8[N0*k]      torch.bool (:781)
 *   rand           f32[N0*k]     the uniform draws of :687, or NULL = keep every candidate
 *   grad_threshold, rand_threshold   the level's compare scalars, already rounded to float32 (torch casts the Python double to the
 *                  tensor's dtype for a compare)
 *   cur_size       voxel_size * size_factor (:704) as the Python DOUBLE: the quotient's reciprocal is taken from it before rounding
 *                  (see LIDARGS_AG_EXACT_DIVISION); new_anchor = float32(voxel) * float32(cur_size)
 *   alloc_work     called at most once: working memory (hash set, sort buffers), free after the call returns
 *   alloc_anchor / alloc_feat   called at most once each, only when U > 0, with exactly U*3*4 and U*F*4 bytes: the outputs
 *                  new_anchor f32[U,3] (candidate_anchor, :730) and new_feat f32[U,F] (:742), rows in torch.unique's order
 *   counts_host    HOST int[3], may be NULL: candidates (:708), their distinct voxels (:711), new anchors U
 */
int lidargs_anchor_growing_level(int N, int N0, int n_offsets, int feat_dim, const float* anchor, const float* offset, const float* scaling,
                                 const float* anchor_feat, const float* grads, const uint8_t* offset_mask, const float* rand,
                                 float grad_threshold, float rand_threshold, double cur_size, int flags, char* scratch, size_t scratch_bytes,
                                 lidargs_alloc_fn alloc_work, void* work_user, lidargs_alloc_fn alloc_anchor, void* anchor_user,
                                 lidargs_alloc_fn alloc_feat, void* feat_user, int* counts_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif
