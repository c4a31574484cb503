"""The framework-op form of one level of GaussianModel.anchor_growing (/root/reference/scene/gaussian_model.py:683-745) restated with
the same torch ops in the same order, runnable on any device.  TEST INFRASTRUCTURE ONLY: the comparator of bench.py's
`anchor_growing` workload (the reference's dataflow timed on the same GPU) and, run on the GPU box, the check that the native call's
default quotient convention is the one torch's device kernels use (tests/test_anchor_growing.py).

`scatter_max` (torch_scatter, absent from this image) is Tensor.scatter_reduce(..., "amax", include_self=False): parity of that one
piece is against torch's reduction, not torch_scatter's ("parity unpinned" for it).
"""
from functools import reduce

import torch


def grow_level(anchor, offset, scaling, anchor_feat, grads, offset_mask, rand, cur_threshold, rand_threshold, cur_size, n_offsets, chunked=True):
    """anchor [N,3], offset [N,k,3], scaling = get_scaling [N,6], anchor_feat [N,F]; grads / offset_mask / rand [N0*k] with N0 <= N.
    Returns (candidate_anchor [U,3], new_feat [U,F], (candidates, distinct voxels, U))."""
    k = n_offsets
    candidate_mask = (grads >= cur_threshold)                                                                      # :683
    candidate_mask = torch.logical_and(candidate_mask, offset_mask)                                                # :684
    if rand is not None:
        candidate_mask = torch.logical_and(candidate_mask, rand > rand_threshold)                                  # :687-689
    length_inc = anchor.shape[0] * k - candidate_mask.shape[0]
    if length_inc > 0:
        candidate_mask = torch.cat([candidate_mask, torch.zeros(length_inc, dtype=torch.bool, device=anchor.device)], dim=0)   # :696
    all_xyz = anchor.unsqueeze(dim=1) + offset * scaling[:, :3].unsqueeze(dim=1)                                   # :698
    grid_coords = torch.round(anchor / cur_size).int()                                                             # :706
    selected_xyz = all_xyz.view([-1, 3])[candidate_mask]                                                           # :708
    selected_grid_coords = torch.round(selected_xyz / cur_size).int()                                              # :709
    F = anchor_feat.shape[1]
    if selected_grid_coords.shape[0] == 0:
        return anchor.new_zeros((0, 3)), anchor.new_zeros((0, F)), (0, 0, 0)
    selected_grid_coords_unique, inverse_indices = torch.unique(selected_grid_coords, return_inverse=True, dim=0)  # :711
    if chunked:                                                                                                    # :714-725
        chunk_size = 4096
        max_iters = grid_coords.shape[0] // chunk_size + (1 if grid_coords.shape[0] % chunk_size != 0 else 0)
        remove_duplicates_list = []
        for i in range(max_iters):
            cur = (selected_grid_coords_unique.unsqueeze(1) == grid_coords[i * chunk_size:(i + 1) * chunk_size, :]).all(-1).any(-1).view(-1)
            remove_duplicates_list.append(cur)
        remove_duplicates = reduce(torch.logical_or, remove_duplicates_list)
    else:
        remove_duplicates = (selected_grid_coords_unique.unsqueeze(1) == grid_coords).all(-1).any(-1).view(-1)    # :727
    remove_duplicates = ~remove_duplicates                                                                         # :729
    candidate_anchor = selected_grid_coords_unique[remove_duplicates] * cur_size                                   # :730
    counts = (int(selected_grid_coords.shape[0]), int(selected_grid_coords_unique.shape[0]), int(candidate_anchor.shape[0]))
    if candidate_anchor.shape[0] == 0:
        return anchor.new_zeros((0, 3)), anchor.new_zeros((0, F)), counts
    new_feat = anchor_feat.unsqueeze(dim=1).repeat([1, k, 1]).view([-1, F])[candidate_mask]                        # :740
    index = inverse_indices.unsqueeze(1).expand(-1, new_feat.size(1))
    best = new_feat.new_zeros((selected_grid_coords_unique.shape[0], F)).scatter_reduce(0, index, new_feat, "amax", include_self=False)   # :742 (stand-in)
    return candidate_anchor, best[remove_duplicates], counts
