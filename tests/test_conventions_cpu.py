"""Pins the two conventions the rasterizer oracle rests on to outputs of the reference's OWN Python (tests/golden/
conventions_golden.npz, produced by tests/golden/make_conventions_golden.py executing utils/graphics_utils.py:38-49,
scene/cameras.py:56,:59, utils/general_utils.py:65-113 and scene/gaussian_model.py:33-37):

  * `viewmatrix` = getWorld2View2(R, T).transpose(0, 1): a row-vector ("transposed") world->lidar matrix, translation in ROW 3;
    the oracle's K1 transform (transformPoint4x3, R3/cr/auxiliary.h:94-102) must map a world point p to R^T p + T under it
    and the camera centre to the origin -- and lidargs_scenes.rigid_viewmatrix must build the same kind of matrix;
  * the covariance of (scale, quaternion): R S S R^T with R the (r, x, y, z) quaternion matrix, packed [00, 01, 02, 11, 12, 22] --
    what the oracle's computeCov3D restatement (R3/cr/forward.cu:216-253) stores, and what a `cov3D_precomp` caller hands over.
"""
import os

import numpy as np
import pytest

import lidargs_scenes as sc
from oracle import lgo

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conventions_golden.npz"))
H, W = 16, 512


def _oracle_view_points(pts, V, far=1000):
    """p_view of each point as the oracle's K1 computes it: state arrays `sphere` (= p_view / |p_view|) and `depths` (= |p_view|)."""
    P = pts.shape[0]
    one = np.ones((P, 1), np.float32)
    quat = np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1))
    f = lgo.forward(pts, np.zeros((P, 2), np.float32), one, 0.05 * np.ones((P, 3), np.float32), quat, V, sc.beam_inclinations(H),
                    W, H, far=far)
    return f.array("sphere").reshape(P, 3) * f.array("depths").reshape(P, 1), f.radii


def test_viewmatrix_is_the_transposed_world_to_view_of_the_reference():
    pts = G["view_points"]
    for R, T, V, cc, w2v in zip(G["view_R"], G["view_T"], G["view_world_view_transform"], G["view_camera_center"], G["view_w2v_untransposed"]):
        assert np.array_equal(V, w2v.T)                                   # scene/cameras.py:56: .transpose(0, 1)
        assert np.allclose(V[3, :3], T, atol=1e-6) and np.allclose(V[:3, 3], 0)       # translation lives in row 3
        expect = pts.astype(np.float64) @ R.astype(np.float64) + T       # = R^T p + T  (getWorld2View2 stores R transposed)
        assert np.allclose(w2v[:3, :3].astype(np.float64) @ pts.T.astype(np.float64) + w2v[:3, 3:4], expect.T, atol=1e-4)
        pv, radii = _oracle_view_points(pts, V)
        # the oracle culls what leaves the beam fan (state rows of culled Gaussians stay zero): compare the rows it kept
        kept = radii > 0
        assert kept.sum() >= 3
        assert np.allclose(pv[kept], expect[kept], rtol=2e-6, atol=2e-5), np.abs(pv[kept] - expect[kept]).max()
        # the camera centre of scene/cameras.py:59 is the pre-image of the origin
        back = np.concatenate([cc, [1.0]]) @ V.astype(np.float64)
        assert np.allclose(back[:3], 0, atol=2e-5)
        # mark_visible reads the same matrix the same way (z_view > 0.2, R3/cr/auxiliary.h:190)
        assert np.array_equal(lgo.mark_visible(pts, V), expect[:, 2].astype(np.float32) > 0.2) or \
            (np.abs(expect[:, 2] - 0.2) < 1e-5).any()


def test_scene_generator_builds_the_same_kind_of_matrix():
    """lidargs_scenes.rigid_viewmatrix (every GPU parity case with random_view=True) follows the reference construction:
    getWorld2View2(R_c2w, t).T has R_c2w in its upper-left block and t in row 3."""
    rng = np.random.default_rng(5)
    V = sc.rigid_viewmatrix(rng)
    Rv = V[:3, :3].astype(np.float64)
    assert np.allclose(Rv @ Rv.T, np.eye(3), atol=1e-6) and np.isclose(np.linalg.det(Rv), 1.0, atol=1e-6)
    assert np.allclose(V[:3, 3], 0) and V[3, 3] == 1.0
    pts = G["view_points"]
    pv, radii = _oracle_view_points(pts, V)
    expect = pts.astype(np.float64) @ Rv + V[3, :3].astype(np.float64)
    kept = radii > 0
    assert kept.sum() >= 3 and np.allclose(pv[kept], expect[kept], rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize("mod,key", [(1.0, "cov_mod1"), (1.7, "cov_mod17")])
def test_covariance_matches_the_reference_python(mod, key):
    scales, q, cov = G["cov_scales"], G["cov_rotations"], G[key]
    P = scales.shape[0]
    rng = np.random.default_rng(1)
    # place the Gaussians inside the beam fan so that none is culled (culled rows keep a zero covariance)
    beams = sc.beam_inclinations(H)
    r = rng.uniform(5, 40, P); az = rng.uniform(-3, 3, P); el = rng.uniform(float(beams[2]), float(beams[-3]), P)
    pts = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1).astype(np.float32)
    one = np.ones((P, 1), np.float32)
    col = rng.uniform(0, 1, (P, 2)).astype(np.float32)
    f = lgo.forward(pts, col, one * 0.7, scales, q, np.eye(4, dtype=np.float32), beams, W, H, scale_modifier=mod)
    assert (f.radii > 0).all()
    mine = f.array("cov3D").reshape(P, 6)
    # fp32 on both sides, different operation order (GLM M^T M vs torch R S (R S)^T): a few ulp of the largest entry
    tol = 4e-6 * np.abs(cov).max(axis=1, keepdims=True) + 1e-12
    assert (np.abs(mine - cov) <= tol).all(), np.abs(mine - cov).max()
    # handing the reference's covariance over as cov3D_precomp renders the same image as scales + rotations
    g = lgo.forward(pts, col, one * 0.7, None, None, np.eye(4, dtype=np.float32), beams, W, H, cov3D_precomp=cov)
    assert (g.radii == f.radii).mean() > 0.99
    assert np.allclose(g.color, f.color, rtol=2e-3, atol=2e-4) and np.allclose(g.depth, f.depth, rtol=2e-3, atol=2e-3)


def test_rotation_matrix_of_a_quaternion():
    """build_rotation (utils/general_utils.py:79-101) on unit quaternions = the matrix whose columns preprocess.hip / the oracle
    expand (r, x, y, z order, no re-normalisation inside the rasterizer, R3/cr/forward.cu:228)."""
    q, Rm = G["cov_rotations"].astype(np.float64), G["cov_rotation_matrices"]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    mine = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], 1),
                     np.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], 1),
                     np.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1)], 1)
    assert np.allclose(mine, Rm, atol=2e-6)
