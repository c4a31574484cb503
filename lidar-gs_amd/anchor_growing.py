"""`GaussianModel.anchor_growing` / `adjust_anchor` of LiDAR-GS (scene/gaussian_model.py:677-775, :776-830) on the native call of
include/lidargs_anchor_growing.h.

    from anchor_growing import anchor_growing, adjust_anchor
    adjust_anchor(gaussians, check_interval=..., success_threshold=..., grad_threshold=..., min_opacity=...)     # train.py:247

Functions of the model (like neural_gaussians.training_statis): the same arguments as the methods, the same attributes read and
replaced (`_anchor`, `_offset`, `_anchor_feat`, `_opacity`, `_scaling`, `_rotation` through the model's own
`cat_tensors_to_optimizer`; `anchor_demon`, `opacity_accum`), the same consumption of the device's random generator (one
`torch.rand_like` of N0*k floats per level, drawn whether the level runs or not, :687), so a run seeded like the reference's grows the
same anchors in the same row order.  Per level the mask, the voxel sets, the duplicate removal against ALL existing anchors and the
per-voxel feature maximum are one native call; the constant fills of the new rows (:734-747) stay framework ops.

There is no CPU path: tensors must live on a HIP device.
"""
import ctypes as C

import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
_lib.lidargs_anchor_growing_level.restype = C.c_int
_lib.lidargs_ag_scratch_bytes.restype = C.c_size_t
EXACT_DIVISION = 1     # LIDARGS_AG_EXACT_DIVISION


def _f32c(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()


def grow_level(anchor, offset, scaling, anchor_feat, grads, offset_mask, rand, grad_threshold, rand_threshold, cur_size, n_initial=None, flags=0):
    """One level (the loop body :683-745 up to the constant fills).  anchor [N,3], offset [N,k,3], scaling = get_scaling [N,6],
    anchor_feat [N,F]; grads / offset_mask / rand cover the first `n_initial` anchors' offsets (default: all N).  The three scalars are
    Python floats: the two thresholds are rounded to float32 (as torch does for a compare), cur_size travels as a double (torch's device
    kernel multiplies by float32(1.0 / cur_size)).
    Returns (new_anchor [U,3], new_feat [U,F], (candidates, distinct voxels, U))."""
    _base._require_device(anchor, "anchor")
    dev = anchor.device
    N, k, F = int(anchor.shape[0]), int(offset.shape[1]), int(anchor_feat.shape[1])
    N0 = N if n_initial is None else int(n_initial)
    anchor, offset, scaling, anchor_feat, grads = _f32c(anchor), _f32c(offset), _f32c(scaling), _f32c(anchor_feat), _f32c(grads).view(-1)
    if offset.shape[0] != N or scaling.shape[0] != N or scaling.shape[1] < 3 or anchor_feat.shape[0] != N:
        raise RuntimeError("anchor_growing: offset / scaling / anchor_feat must have one row per anchor")
    if scaling.shape[1] != 6:
        scaling = torch.cat([scaling[:, :3], scaling[:, :3]], 1).contiguous()
    om = offset_mask.detach().to(torch.bool).contiguous().view(-1).view(torch.uint8)
    if grads.numel() != N0 * k or om.numel() != N0 * k or (rand is not None and rand.numel() != N0 * k):
        raise RuntimeError("anchor_growing: grads / offset_mask / rand must have n_initial * n_offsets entries")
    rnd = None if rand is None else _f32c(rand).view(-1)
    empty = (torch.empty((0, 3), dtype=torch.float32, device=dev), torch.empty((0, F), dtype=torch.float32, device=dev))
    if N0 == 0:
        return empty + ((0, 0, 0),)
    nb = int(_lib.lidargs_ag_scratch_bytes(C.c_int(N0), C.c_int(k)))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    work, out_a, out_f = _base._Scratch(dev), _base._Scratch(dev), _base._Scratch(dev)
    counts = (C.c_int * 3)()
    p = _base._ptr
    with torch.cuda.device(dev):
        rc = _lib.lidargs_anchor_growing_level(C.c_int(N), C.c_int(N0), C.c_int(k), C.c_int(F), p(anchor), p(offset), p(scaling), p(anchor_feat), p(grads), p(om),
                                               p(rnd), C.c_float(grad_threshold), C.c_float(rand_threshold), C.c_double(cur_size), C.c_int(flags), p(scratch),
                                               C.c_size_t(nb), work.cb, work.user, out_a.cb, out_a.user, out_f.cb, out_f.user, counts, _base._stream(dev))
    work.take()
    ta, tf = out_a.take(), out_f.take()
    if rc < 0:
        _base._raise(rc, "lidargs_anchor_growing_level")
    if rc == 0:
        return empty + (tuple(counts),)
    return ta.view(torch.float32).view(rc, 3), tf.view(torch.float32).view(rc, F), tuple(counts)


def anchor_growing(pc, grads, threshold, offset_mask, flags=0):
    """Drop-in for GaussianModel.anchor_growing(grads, threshold, offset_mask), called as anchor_growing(gaussians, ...)."""
    k = int(pc.n_offsets)
    n_initial = int(pc.get_anchor.shape[0])
    dev = pc.get_anchor.device
    with torch.no_grad():
        for i in range(pc.update_depth):
            cur_threshold = threshold * ((pc.update_hierachy_factor // 2) ** i)                                    # :682
            rand = torch.rand_like(grads.detach().view(-1).float())                                                # :687 (drawn before the level may be skipped)
            N = int(pc.get_anchor.shape[0])
            if N == n_initial and i > 0:                                                                           # :690-693
                continue
            cur_size = pc.voxel_size * (pc.update_init_factor // (pc.update_hierachy_factor ** i))                 # :703-704
            new_anchor, new_feat, _ = grow_level(pc.get_anchor, pc._offset, pc.get_scaling, pc._anchor_feat, grads, offset_mask, rand, cur_threshold,
                                                 0.5 ** (i + 1), cur_size, n_initial=n_initial, flags=flags)
            U = int(new_anchor.shape[0])
            if U == 0:                                                                                             # :733
                continue
            new_scaling = torch.log(torch.ones((U, 6), dtype=torch.float32, device=dev) * cur_size)                # :734-735
            new_rotation = torch.zeros((U, 4), dtype=torch.float32, device=dev)
            new_rotation[:, 0] = 1.0                                                                               # :736-737
            x = 0.9 * torch.ones((U, 1), dtype=torch.float32, device=dev)
            new_opacities = torch.log(x / (1 - x))                                                                 # :738 inverse_sigmoid(0.9), utils/general_utils.py:19-20
            d = {"anchor": new_anchor, "scaling": new_scaling, "rotation": new_rotation, "anchor_feat": new_feat,
                 "offset": torch.zeros((U, k, 3), dtype=torch.float32, device=dev), "opacity": new_opacities}     # :747-756
            zeros = torch.zeros((U, 1), dtype=torch.float32, device=dev)
            pc.anchor_demon = torch.cat([pc.anchor_demon, zeros], dim=0)                                           # :759-761
            pc.opacity_accum = torch.cat([pc.opacity_accum, zeros], dim=0)                                         # :763-765
            t = pc.cat_tensors_to_optimizer(d)                                                                     # :769 (the model's own optimizer surgery)
            pc._anchor, pc._scaling, pc._rotation = t["anchor"], t["scaling"], t["rotation"]
            pc._anchor_feat, pc._offset, pc._opacity = t["anchor_feat"], t["offset"], t["opacity"]


def adjust_anchor(pc, check_interval=100, success_threshold=0.8, grad_threshold=0.0002, min_opacity=0.005, flags=0):
    """Drop-in for GaussianModel.adjust_anchor (:776-830): the growing through the native call, the bookkeeping around it (statistics
    reset / padding, the prune masks and the model's own prune_anchor) as the reference writes it."""
    with torch.no_grad():
        grads = pc.offset_gradient_accum / pc.offset_denom                                                         # :778
        grads[grads.isnan()] = 0.0
        grads_norm = torch.norm(grads, dim=-1)
        offset_mask = (pc.offset_denom > check_interval * success_threshold).squeeze(dim=1)                        # :781
        anchor_growing(pc, grads_norm, grad_threshold, offset_mask, flags=flags)
        k = int(pc.n_offsets)
        dev = pc.offset_denom.device
        pc.offset_denom[offset_mask] = 0                                                                           # :786
        pad = pc.get_anchor.shape[0] * k - pc.offset_denom.shape[0]
        pc.offset_denom = torch.cat([pc.offset_denom, torch.zeros([pad, 1], dtype=torch.int32, device=dev)], dim=0)
        pc.offset_gradient_accum[offset_mask] = 0                                                                  # :792
        pc.offset_gradient_accum = torch.cat([pc.offset_gradient_accum, torch.zeros([pad, 1], dtype=torch.int32, device=dev)], dim=0)
        prune_mask = (pc.opacity_accum < min_opacity * pc.anchor_demon).squeeze(dim=1)                             # :799-801
        anchors_mask = (pc.anchor_demon > check_interval * success_threshold).squeeze(dim=1)
        prune_mask = torch.logical_and(prune_mask, anchors_mask)
        pc.offset_denom = pc.offset_denom.view([-1, k])[~prune_mask].view([-1, 1])                                # :804-812
        pc.offset_gradient_accum = pc.offset_gradient_accum.view([-1, k])[~prune_mask].view([-1, 1])
        if anchors_mask.sum() > 0:                                                                                 # :815-817
            pc.opacity_accum[anchors_mask] = 0.0
            pc.anchor_demon[anchors_mask] = 0.0
        pc.opacity_accum = pc.opacity_accum[~prune_mask]                                                           # :819-825
        pc.anchor_demon = pc.anchor_demon[~prune_mask]
        if prune_mask.shape[0] > 0:                                                                                # :827-828
            pc.prune_anchor(prune_mask)
        pc.max_radii2D = torch.zeros((pc.get_anchor.shape[0]), device=dev)                                         # :830
