"""CPU checks of the surfel oracle (oracle/lidargs_surfel_oracle.c, restating R2): output invariants the reference's
algorithm implies, a finite-difference check of its analytic backward, the summation-order band, and the filter kernel."""
import numpy as np
import pytest

from oracle import lgo, lgo_surfel
from util import GRAD_KEYS_SURFEL, surfel_scene, surfel_upstream_grads

W, H, P, SEED = 512, 16, 3000, 3


def _fwd(sc, **kw):
    return lgo_surfel.forward(sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], sc["viewmatrix"],
                              sc["beams"], W, H, bg=sc["bg"], **kw)


@pytest.fixture(scope="module")
def shell():
    sc = surfel_scene("shell", P, H, SEED)
    return sc, _fwd(sc)


def test_output_invariants(shell):
    sc, f = shell
    depth, alpha, nrm, med, dist = f.others[0], f.others[1], f.others[2:5], f.others[5], f.others[6]
    T = f.array("accum")[:W * H].reshape(H, W)
    np.testing.assert_allclose(alpha, 1.0 - T, atol=1e-6)                       # others[1] = 1 - T (R2/cr/forward.cu:536)
    assert (alpha >= 0).all() and (alpha <= 1.0).all()
    assert (np.linalg.norm(nrm, axis=0) <= alpha + 1e-5).all()                  # sum of w_i n_i with unit normals
    hit = alpha > 0
    assert (depth[hit] > 0).all() and (depth[~hit] == 0).all()
    assert ((med == 0) | (med >= 0.2)).all()                                    # median depth is one of the blended depths (> near_n)
    assert (np.abs(dist) < 1e-2).all()
    assert f.num_rendered == int(f.array("tiles_touched").sum()) > 0
    assert ((f.radii > 0) == (f.array("tiles_touched") > 0)).all()
    # colour = sum w c + T bg with bg = 0 here: bounded by alpha * max colour
    assert (np.abs(f.color) <= alpha[None] * np.abs(sc["colors"]).max() + 1e-5).all()


def test_normals_face_the_sensor(shell):
    sc, f = shell
    no = f.array("normal_opacity").reshape(-1, 4)
    tm = f.array("transMat").reshape(-1, 9)
    vis = f.radii > 0
    cosang = -(tm[vis, 6:9] * no[vis, :3]).sum(1)                               # DUAL_VISIABLE flip (R2/cr/forward.cu:297-302)
    assert (cosang > 0).all()
    np.testing.assert_allclose(np.linalg.norm(no[vis, :3], axis=1), 1.0, atol=1e-5)


def test_background_and_range_window():
    sc = surfel_scene("shell", 1000, H, 8)
    sc["bg"] = np.array([0.3, -0.2], np.float32)
    f = _fwd(sc)
    T = f.array("accum")[:W * H].reshape(H, W)
    empty = T == 1.0
    assert empty.any()
    np.testing.assert_allclose(f.color[0][empty], 0.3, atol=1e-7)
    np.testing.assert_allclose(f.color[1][empty], -0.2, atol=1e-7)
    near = _fwd(sc, far=20)
    tm = f.array("transMat").reshape(-1, 9)
    assert (np.linalg.norm(tm[near.radii > 0, 6:9], axis=1) < 20).all()


def test_backward_matches_finite_differences():
    """Colours and opacities enter linearly / smoothly: the analytic gradients must match central differences of
    L = <g_color, color> + <g_others[0,1,2:5], depth/alpha/normal> (median and distortion excluded: the first is a
    selection, the second's weight gradient is detached by the reference, DETACH_WEIGHT R2/cr/auxiliary.h:31)."""
    sc = surfel_scene("shell", 300, H, 12)
    gc, go = surfel_upstream_grads(H, W, 12)
    go[5:] = 0
    f = _fwd(sc)
    g = lgo_surfel.backward(f, gc, go)

    def loss(s):
        ff = _fwd(s)
        return float((gc.astype(np.float64) * ff.color).sum() + (go[:5].astype(np.float64) * ff.others[:5]).sum())
    rng = np.random.default_rng(0)
    vis = np.nonzero(f.radii > 0)[0]
    checked = 0
    for i in rng.choice(vis, 12, replace=False):
        for key, gkey, col, eps in (("colors", "dL_dcolors", 1, 1e-2), ("colors", "dL_dcolors", 0, 1e-2)):
            sp, sm = {k: v.copy() for k, v in sc.items()}, {k: v.copy() for k, v in sc.items()}
            sp[key][i, col] += eps; sm[key][i, col] -= eps
            fd = (loss(sp) - loss(sm)) / (2 * eps)
            an = g[gkey][i, col]
            if col == 0:
                # channel 0's dL/dalpha path exists, but d colour/d colour is exact for both channels
                pass
            assert abs(fd - an) <= 2e-3 * max(1.0, abs(an)), (i, key, col, fd, an)
            checked += 1
    assert checked == 24


def test_transmat_precomp_replaces_the_blend_rows(shell):
    """transMat_precomp (R2/cr/rasterizer_impl.cu:332, :408): the rows the preprocess builds, handed back in, change nothing; other rows
    change the image but not the rects, and the backward's per-pair terms with it (no finite-difference check on the rows: central
    differences of the colour loss on a Tu component gave 0.09 against an analytic 0.55 -- the restated backward carries the reference's
    detached weights and its 2-D / 3-D branch selection, so, as in the test above, only inputs that enter linearly are checked that way)."""
    sc, f = shell
    tm = f.array("transMat").reshape(-1, 9)
    same = _fwd(sc, transMat_precomp=tm)
    assert np.array_equal(same.color, f.color) and np.array_equal(same.others, f.others) and np.array_equal(same.radii, f.radii)
    rng = np.random.default_rng(5)
    tm2 = (tm * (1.0 + 0.05 * rng.normal(size=tm.shape))).astype(np.float32)
    pert = _fwd(sc, transMat_precomp=tm2)
    assert np.array_equal(pert.radii, f.radii) and pert.num_rendered == f.num_rendered
    assert np.abs(pert.color - f.color).max() > 1e-3
    # the backward reads the same rows: with the preprocess's own rows handed back in, every gradient is the plain one, bit for bit
    gc, go = surfel_upstream_grads(H, W, 12)
    g0, g1 = lgo_surfel.backward(f, gc, go), lgo_surfel.backward(same, gc, go)
    for k in GRAD_KEYS_SURFEL + ("dL_dtransMat",):
        assert np.array_equal(g0[k], g1[k]), k
    # ... and with other rows the per-pair gradients move while culled surfels keep zero rows
    g2 = lgo_surfel.backward(pert, gc, go)
    assert np.abs(g2["dL_dtransMat"] - g0["dL_dtransMat"]).max() > 1e-3
    assert (g2["dL_dtransMat"][f.radii == 0] == 0).all()


def test_summation_order_band(shell):
    sc, f = shell
    g = surfel_upstream_grads(H, W, SEED)
    a = lgo_surfel.backward(f, *g)
    lgo.lib().sfo_set_reverse_pixel_order(1)
    try:
        b = lgo_surfel.backward(f, *g)
    finally:
        lgo.lib().sfo_set_reverse_pixel_order(0)
    for k in GRAD_KEYS_SURFEL:
        scale = np.abs(a[k]).max()
        err = np.abs(a[k] - b[k]) / (np.abs(a[k]) + 1e-3 * scale)
        assert err.max() < 5e-5, (k, err.max())


def test_culled_rows_have_zero_gradients(shell):
    sc, f = shell
    g = lgo_surfel.backward(f, *surfel_upstream_grads(H, W, SEED))
    dead = f.radii == 0
    assert dead.any()
    for k in GRAD_KEYS_SURFEL + ("dL_dtransMat", "dL_dnormal"):
        assert (g[k][dead] == 0).all(), k
    assert np.isfinite(np.concatenate([g[k].ravel() for k in GRAD_KEYS_SURFEL])).all()


def test_visible_filter_matches_forward_radii(shell):
    sc, f = shell
    r = lgo_surfel.visible_filter(sc["means3D"], sc["scales"], sc["rotations"], sc["viewmatrix"], sc["beams"], W, H)
    # the filter kernel skips the degenerate-normal test of the forward (R2/cr/forward.cu:551-631 vs :297-302)
    assert (r != f.radii).sum() <= 1


def test_empty_input():
    sc = surfel_scene("shell", 10, H, 1)
    e = {k: (v[:0] if k in ("means3D", "colors", "opacities", "scales", "rotations") else v) for k, v in sc.items()}
    f = _fwd(e)
    assert f.num_rendered == 0 and f.color.shape == (2, H, W) and (f.others == 0).all()
