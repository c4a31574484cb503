"""CPU restatement (numpy) of GaussianModel.anchor_growing (/root/reference/scene/gaussian_model.py:677-775), SURVEY section 8 row
f4 (second half): growing new anchors where the accumulated view-space gradient of an anchor's offsets is large.
TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg).

PARITY STATUS: pinned against tests/golden/anchor_growing_golden.npz, produced by EXECUTING the reference method on CPU torch
(tests/golden/make_anchor_growing_golden.py) -- EXCEPT `scatter_max`: torch_scatter is not in this image, the generator substitutes
`Tensor.scatter_reduce(..., "amax", include_self=False)` and says so; that one piece (the per-voxel feature maximum, :745) is
"parity unpinned" against torch_scatter's own kernel (a maximum has one answer up to the sign of zero).

Two conventions of `tensor / python_float` exist in torch and both are restated (`exact_division`):
  True   IEEE division            -- torch's CPU kernel, hence what the fixture pins;
  False  x * float32(1.0 / size)  -- torch's device kernel (ATen BinaryDivTrueKernel.cu: "if the second operand is a CPU scalar, compute
                                     a * reciprocal(b)"; the reciprocal of the Python double, then rounded: measured against torch-ROCm on
                                     the MI355X, profiles/r06_div_convention.txt), hence what the reference computes where it really runs
                                     (every tensor is .cuda()).
"""
import numpy as np

F32 = np.float32


def _quantize(x, cur_size, exact_division):
    """torch.round(x / cur_size).int()  (:706, :709): float32 quotient, round half to even, cast."""
    s = F32(cur_size)
    q = (x / s) if exact_division else (x * F32(1.0 / float(cur_size)))
    return np.rint(q.astype(F32)).astype(np.int32)


def _rows_in(rows, table):
    """bool[len(rows)]: is the int32 row one of `table`'s rows?  Rows as one int64 each when the joint box allows, Python tuples otherwise."""
    lo = np.minimum(rows.min(0), table.min(0)).astype(np.int64)
    ext = np.maximum(rows.max(0), table.max(0)).astype(np.int64) - lo + 1
    if float(ext[0]) * float(ext[1]) * float(ext[2]) < 2.0 ** 62:
        one = lambda g: ((g[:, 0].astype(np.int64) - lo[0]) * ext[1] + (g[:, 1].astype(np.int64) - lo[1])) * ext[2] + (g[:, 2].astype(np.int64) - lo[2])
        return np.isin(one(rows), one(table))
    taken = set(map(tuple, table.tolist()))
    return np.array([tuple(r) in taken for r in rows.tolist()], bool)


def level_parameters(level, threshold, voxel_size, update_init_factor=16, update_hierachy_factor=4):
    """(gradient threshold, random threshold, voxel edge) of loop iteration `level` (:682, :688, :703-704) as Python floats."""
    cur_threshold = threshold * ((update_hierachy_factor // 2) ** level)
    size_factor = update_init_factor // (update_hierachy_factor ** level)
    return cur_threshold, 0.5 ** (level + 1), voxel_size * size_factor


def candidate_mask(grads, offset_mask, rand, cur_threshold, rand_threshold, n_slots):
    """:684-696: float32 compares (a Python scalar is cast to the tensor's dtype), padded with False for anchors grown since."""
    m = (grads.astype(F32) >= F32(cur_threshold)) & np.asarray(offset_mask, bool)
    if rand is not None:
        m &= rand.astype(F32) > F32(rand_threshold)
    out = np.zeros(n_slots, bool)
    out[:m.shape[0]] = m
    return out


def grow_level(anchor, offset, scaling, anchor_feat, cand, cur_size, exact_division=True):
    """One iteration of the loop body, :698-745.  anchor f32[N,3], offset f32[N,k,3], scaling = get_scaling f32[N,6] (exp-activated),
    anchor_feat f32[N,F], cand bool[N*k].  Returns (new_anchor f32[U,3], new_feat f32[U,F], n_candidates, n_voxels)."""
    N, k = offset.shape[0], offset.shape[1]
    all_xyz = anchor[:, None, :].astype(F32) + (offset.astype(F32) * scaling[:, None, :3].astype(F32)).astype(F32)        # :698 (two roundings)
    grid = _quantize(anchor.astype(F32), cur_size, exact_division)                                                            # :706
    sel = all_xyz.reshape(-1, 3)[cand]                                                                                        # :708
    sel_grid = _quantize(sel, cur_size, exact_division)                                                                       # :709
    if sel_grid.shape[0] == 0:
        return np.zeros((0, 3), F32), np.zeros((0, anchor_feat.shape[1]), F32), 0, 0
    uniq, inverse = np.unique(sel_grid, axis=0, return_inverse=True)                                                          # :711 (sorted rows)
    inverse = inverse.reshape(-1)
    # :714-727: is the voxel already an anchor's?  (the reference compares every pair in chunks of 4096 anchors)
    keep = ~_rows_in(uniq, grid)                                                                                             # :729
    new_anchor = (uniq[keep].astype(F32) * F32(cur_size)).astype(F32)                                                         # :730
    feat = np.repeat(anchor_feat.astype(F32), k, axis=0)[cand]                                                                # :740
    best = np.full((uniq.shape[0], anchor_feat.shape[1]), -np.inf, F32)
    np.maximum.at(best, inverse, feat)                                                                                        # :742 scatter_max
    return new_anchor, best[keep], int(sel_grid.shape[0]), int(uniq.shape[0])


def anchor_growing(state, grads, threshold, offset_mask, rands, voxel_size, update_depth=3, update_init_factor=16, update_hierachy_factor=4,
                   exact_division=True):
    """The whole method on a dict of arrays (anchor, offset, scaling [raw, log-space], anchor_feat, rotation, opacity, anchor_demon,
    opacity_accum); `rands` = the update_depth arrays torch.rand_like returned (:687).  Returns the grown copies and the per-level counts."""
    s = {key: np.array(v, copy=True) for key, v in state.items()}
    k = s["offset"].shape[1]
    init_len = s["anchor"].shape[0] * k
    counts = []
    for i in range(update_depth):
        thr, rthr, cur_size = level_parameters(i, threshold, voxel_size, update_init_factor, update_hierachy_factor)
        length_inc = s["anchor"].shape[0] * k - init_len
        if length_inc == 0 and i > 0:                                                                                         # :691-693 (levels > 0 only run once level 0 grew something)
            counts.append((-1, -1, -1))
            continue
        cand = candidate_mask(grads, offset_mask, rands[i], thr, rthr, s["anchor"].shape[0] * k)
        new_anchor, new_feat, n_c, n_v = grow_level(s["anchor"], s["offset"], np.exp(s["scaling"].astype(F32)).astype(F32), s["anchor_feat"], cand, cur_size, exact_division)
        U = new_anchor.shape[0]
        counts.append((n_c, n_v, U))
        if U == 0:                                                                                                            # :733
            continue
        one = np.ones((U, 1), F32)
        s["anchor"] = np.concatenate([s["anchor"], new_anchor])
        s["scaling"] = np.concatenate([s["scaling"], np.log(np.ones((U, 6), F32) * F32(cur_size)).astype(F32)])                # :734-735
        rot = np.zeros((U, 4), F32); rot[:, 0] = 1                                                                            # :736-737
        s["rotation"] = np.concatenate([s["rotation"], rot])
        x = F32(0.9) * one
        s["opacity"] = np.concatenate([s["opacity"], np.log(x / (F32(1) - x)).astype(F32)])                                   # :738 inverse_sigmoid(0.9)
        s["anchor_feat"] = np.concatenate([s["anchor_feat"], new_feat])
        s["offset"] = np.concatenate([s["offset"], np.zeros((U, k, 3), F32)])                                                 # :747
        s["anchor_demon"] = np.concatenate([s["anchor_demon"], 0 * one])                                                      # :759
        s["opacity_accum"] = np.concatenate([s["opacity_accum"], 0 * one])                                                    # :763
    return s, counts
