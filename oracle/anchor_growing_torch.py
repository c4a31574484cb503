"""The framework-op form of one level of GaussianModel.anchor_growing (/root/reference/scene/gaussian_model.py:683-745): the same torch
ops in the same order on any device, written out again with this repository's names.  TEST INFRASTRUCTURE ONLY: the comparator of
bench.py's `anchor_growing` workload (the reference's dataflow timed on the same GPU) and, run on the GPU box, the check that the native
call's default quotient convention is the one torch's device kernels use (tests/test_anchor_growing.py).

`scatter_max` (torch_scatter, absent from this image) is Tensor.scatter_reduce(..., "amax", include_self=False): parity of that one
piece is against torch's reduction, not torch_scatter's ("parity unpinned" for it).
"""
import torch

PAIR_SCAN_CHUNK = 4096      # anchors per slice of the (candidate voxel x anchor) comparison, :716


def grow_level(anchor, offset, scaling, anchor_feat, grads, offset_mask, rand, grad_threshold, rand_threshold, voxel_edge, n_offsets, chunked=True):
    """anchor [N,3], offset [N,k,3], scaling = get_scaling [N,6], anchor_feat [N,F]; grads / offset_mask / rand [N0*k] with N0 <= N.
    Returns (new anchors [U,3], their features [U,F], (candidates, distinct voxels, U))."""
    k, F, dev = n_offsets, anchor_feat.shape[1], anchor.device
    cand = torch.logical_and(grads >= grad_threshold, offset_mask)                                      # :683-684
    if rand is not None:
        cand = torch.logical_and(cand, rand > rand_threshold)                                           # :687-689
    grown = anchor.shape[0] * k - cand.shape[0]
    if grown > 0:                                                                                       # :696 anchors added by earlier levels have no candidates
        cand = torch.cat([cand, torch.zeros(grown, dtype=torch.bool, device=dev)], dim=0)
    positions = anchor.unsqueeze(dim=1) + offset * scaling[:, :3].unsqueeze(dim=1)                      # :698
    anchor_voxels = torch.round(anchor / voxel_edge).int()                                              # :706
    cand_voxels = torch.round(positions.view([-1, 3])[cand] / voxel_edge).int()                         # :708-709
    none = (anchor.new_zeros((0, 3)), anchor.new_zeros((0, F)))
    if cand_voxels.shape[0] == 0:
        return none + ((0, 0, 0),)
    distinct, which = torch.unique(cand_voxels, return_inverse=True, dim=0)                             # :711
    if chunked:                                                                                         # :714-725: every (distinct voxel, anchor) pair, a slice of anchors at a time
        hits = None
        for lo in range(0, anchor_voxels.shape[0], PAIR_SCAN_CHUNK):
            hit = (distinct.unsqueeze(1) == anchor_voxels[lo:lo + PAIR_SCAN_CHUNK, :]).all(-1).any(-1).view(-1)
            hits = hit if hits is None else torch.logical_or(hits, hit)
    else:
        hits = (distinct.unsqueeze(1) == anchor_voxels).all(-1).any(-1).view(-1)                        # :727
    free = ~hits                                                                                        # :729
    new_anchor = distinct[free] * voxel_edge                                                            # :730
    counts = (int(cand_voxels.shape[0]), int(distinct.shape[0]), int(new_anchor.shape[0]))
    if new_anchor.shape[0] == 0:
        return none + (counts,)
    feats = anchor_feat.unsqueeze(dim=1).repeat([1, k, 1]).view([-1, F])[cand]                          # :740
    index = which.unsqueeze(1).expand(-1, F)
    best = feats.new_zeros((distinct.shape[0], F)).scatter_reduce(0, index, feats, "amax", include_self=False)   # :742 (stand-in for scatter_max)
    return new_anchor, best[free], counts
