"""Row f4: densification statistics (GaussianModel.training_statis).  CPU: the numpy oracle against golden vectors from executing the
reference method.  GPU: the fused native call against the same golden vectors and the oracle on a larger case."""
import os
import types

import numpy as np
import pytest

from oracle import training_statis as ots

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "training_statis_golden.npz")
FIELDS = ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")


def load(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_matches_reference_execution(tag):
    c = load(tag)
    out = ots.training_statis({f: c["before_" + f] for f in FIELDS}, c["grad"], c["opacity"], c["update_filter"], c["sel"], c["vis"], int(c["k"]))
    for f in FIELDS:
        np.testing.assert_allclose(out[f], c["after_" + f], rtol=1e-6, atol=1e-7, err_msg=f)


def _run_hip(c, k):
    import torch
    from neural_gaussians import training_statis
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pc = types.SimpleNamespace(n_offsets=k, **{f: t(c["before_" + f]) for f in FIELDS})
    training_statis(pc, types.SimpleNamespace(grad=t(c["grad"])), t(c["opacity"]), t(c["update_filter"]), t(c["sel"]), t(c["vis"]))
    return {f: getattr(pc, f).cpu().numpy() for f in FIELDS}


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_matches_reference_golden(tag, hip_lib_built):
    c = load(tag)
    out = _run_hip(c, int(c["k"]))
    for f in FIELDS:
        np.testing.assert_allclose(out[f], c["after_" + f], rtol=2e-6, atol=1e-6, err_msg=f)


@pytest.mark.gpu
def test_hip_matches_oracle_large_and_degenerate(hip_lib_built):
    rng = np.random.default_rng(8)
    for N, k, pvis in ((200_001, 6, 0.7), (5000, 10, 0.0), (5000, 4, 1.0)):
        vis = rng.random(N) < pvis
        n = int(vis.sum())
        opacity = (rng.random((n * k, 1)) * 2 - 1).astype(np.float32)
        sel = (opacity > 0).reshape(-1)
        M = int(sel.sum())
        c = dict(vis=vis, opacity=opacity, sel=sel, update_filter=rng.random(M) > 0.5, grad=rng.normal(size=(M, 4)).astype(np.float32))
        for f, shape in zip(FIELDS, ((N, 1), (N, 1), (N * k, 1), (N * k, 1))):
            c["before_" + f] = rng.random(shape).astype(np.float32)
        ref = ots.training_statis({f: c["before_" + f] for f in FIELDS}, c["grad"], opacity, c["update_filter"], sel, vis, k)
        out = _run_hip(c, k)
        for f in FIELDS:
            np.testing.assert_allclose(out[f], ref[f], rtol=2e-6, atol=1e-6, err_msg=f"{f} N={N}")
