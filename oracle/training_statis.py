"""CPU restatement (numpy) of GaussianModel.training_statis (/root/reference/scene/gaussian_model.py:599-622), SURVEY section 8
row f4: the per-iteration densification statistics.  TEST INFRASTRUCTURE ONLY.  PARITY STATUS: pinned against
tests/golden/training_statis_golden.npz (produced by executing the reference method, tests/golden/make_training_statis_golden.py)."""
import numpy as np


def training_statis(state, grad, opacity, update_filter, offset_selection_mask, anchor_visible_mask, k):
    """state: dict of opacity_accum [N,1], anchor_demon [N,1], offset_gradient_accum [N*k,1], offset_denom [N*k,1] (updated copies
    are returned).  grad [M,4] = viewspace_point_tensor.grad; opacity [n*k,1]; update_filter bool[M]; offset_selection_mask
    bool[n*k]; anchor_visible_mask bool[N]."""
    s = {key: v.copy() for key, v in state.items()}
    vis = np.asarray(anchor_visible_mask, bool)
    t = np.maximum(opacity.reshape(-1), 0).reshape(-1, k)                                   # :601-604
    s["opacity_accum"][vis] += t.sum(1, keepdims=True).astype(np.float32)                  # :605
    s["anchor_demon"][vis] += 1                                                             # :608
    combined = np.zeros(s["offset_gradient_accum"].shape[0], bool)
    combined[np.repeat(vis, k)] = np.asarray(offset_selection_mask, bool)                   # :611-613
    tmp = combined.copy()
    combined[tmp] = np.asarray(update_filter, bool)                                         # :614-615
    g = grad[np.asarray(update_filter, bool)][:, 2:]
    norm = np.sqrt((g.astype(np.float32) ** 2).sum(1, keepdims=True, dtype=np.float32)).astype(np.float32)   # :618
    s["offset_gradient_accum"][combined] += norm                                            # :619
    s["offset_denom"][combined] += 1                                                        # :620
    return s
