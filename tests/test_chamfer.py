"""Row f3: the chamfer nearest-neighbour kernel.  CPU: the oracle against scipy's KD-tree.  GPU: the `chamfer_3D` drop-in module
(C ABI lidargs_chamfer_*) against the oracle -- distances within 1e-6 relative, indices bit-exact."""
import numpy as np
import pytest

from oracle import chamfer3d


def clouds(B, n, m, seed, scale=20.0):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(B, n, 3)) * scale).astype(np.float32), (rng.normal(size=(B, m, 3)) * scale).astype(np.float32)


def test_oracle_agrees_with_kdtree():
    from scipy.spatial import cKDTree
    a, b = clouds(1, 1500, 2100, 1)
    d1, d2, i1, i2 = chamfer3d.forward(a, b)
    dd, ii = cKDTree(b[0].astype(np.float64)).query(a[0].astype(np.float64))
    assert (i1[0] == ii).mean() > 0.999                                 # a float32 near-tie may pick the other neighbour
    np.testing.assert_allclose(d1[0], dd ** 2, rtol=1e-5)
    dd, ii = cKDTree(a[0].astype(np.float64)).query(b[0].astype(np.float64))
    np.testing.assert_allclose(d2[0], dd ** 2, rtol=1e-5)


def test_oracle_ties_take_the_lowest_index_and_backward_is_consistent():
    a = np.zeros((1, 3, 3), np.float32); a[0, 1] = (1, 0, 0); a[0, 2] = (5, 5, 5)
    b = np.array([[[2, 0, 0], [0.5, 0, 0], [0.5, 0, 0], [-0.5, 0, 0]]], np.float32)
    d1, d2, i1, i2 = chamfer3d.forward(a, b)
    assert i1[0].tolist() == [1, 1, 0] and i2[0].tolist() == [1, 0, 0, 0]      # equidistant candidates: first one wins
    g1, g2 = np.ones_like(d1), np.ones_like(d2)
    gx1, gx2 = chamfer3d.backward(a, b, g1, g2, i1, i2)
    eps = 1e-3
    ap = a.copy(); ap[0, 2, 0] += eps
    dp1, dp2, _, _ = chamfer3d.forward(ap, b)
    fd = ((dp1.sum() + dp2.sum()) - (d1.sum() + d2.sum())) / eps
    assert abs(fd - gx1[0, 2, 0]) < 2e-2 * abs(fd)


@pytest.mark.gpu
@pytest.mark.parametrize("B,n,m", [(1, 5000, 7000), (2, 1025, 1023), (1, 1, 3000), (3, 700, 1), (1, 4096, 4096)])
def test_hip_matches_oracle(B, n, m, hip_lib_built):
    import torch
    import chamfer_3D
    a, b = clouds(B, n, m, 10 + n % 7)
    b[:, : min(m, 50)] = a[:, : min(m, 50)][:, :: -1][:, : min(m, 50)] if n >= 50 and m >= 50 else b[:, : min(m, 50)]   # exact coincidences -> distance 0
    ref = chamfer3d.forward(a, b)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d1, d2 = torch.zeros(B, n, device="cuda"), torch.zeros(B, m, device="cuda")
    i1, i2 = torch.zeros(B, n, dtype=torch.int32, device="cuda"), torch.zeros(B, m, dtype=torch.int32, device="cuda")
    chamfer_3D.forward(ta, tb, d1, d2, i1, i2)
    assert np.array_equal(i1.cpu().numpy(), ref[2]) and np.array_equal(i2.cpu().numpy(), ref[3])          # bit-exact indices
    assert np.array_equal(d1.cpu().numpy(), ref[0]) and np.array_equal(d2.cpu().numpy(), ref[1])          # and distances (same fp32 expression)
    rng = np.random.default_rng(5)
    g1, g2 = rng.normal(size=(B, n)).astype(np.float32), rng.normal(size=(B, m)).astype(np.float32)
    gx1, gx2 = torch.zeros_like(ta), torch.zeros_like(tb)
    chamfer_3D.backward(ta, tb, gx1, gx2, torch.from_numpy(g1).cuda(), torch.from_numpy(g2).cuda(), i1, i2)
    r1, r2 = chamfer3d.backward(a, b, g1, g2, ref[2], ref[3])
    for got, want in ((gx1.cpu().numpy(), r1), (gx2.cpu().numpy(), r2)):
        err = np.abs(got - want) / (np.abs(want) + 1e-3 * max(np.abs(want).max(), 1e-30))
        assert err.max() < 1e-4


@pytest.mark.gpu
def test_hip_duplicates_pick_lowest_index_and_errors(hip_lib_built):
    import torch
    import chamfer_3D
    a = torch.zeros(1, 300, 3, device="cuda")
    b = torch.zeros(1, 2000, 3, device="cuda")                        # every target equidistant: index 0 everywhere
    d1, d2 = torch.empty(1, 300, device="cuda"), torch.empty(1, 2000, device="cuda")
    i1, i2 = torch.empty(1, 300, dtype=torch.int32, device="cuda"), torch.empty(1, 2000, dtype=torch.int32, device="cuda")
    chamfer_3D.forward(a, b, d1, d2, i1, i2)
    assert int(i1.abs().sum()) == 0 and int(i2.abs().sum()) == 0 and float(d1.sum()) == 0.0
    with pytest.raises(RuntimeError, match="HIP device"):
        chamfer_3D.forward(a.cpu(), b, d1, d2, i1, i2)
    with pytest.raises(RuntimeError, match="int32"):
        chamfer_3D.forward(a, b, d1, d2, i1.long(), i2)


def _hip_forward(a, b):
    import torch
    import chamfer_3D
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d1, d2 = torch.full((B, n), -1.0, device="cuda"), torch.full((B, m), -1.0, device="cuda")
    i1, i2 = torch.full((B, n), -1, dtype=torch.int32, device="cuda"), torch.full((B, m), -1, dtype=torch.int32, device="cuda")
    chamfer_3D.forward(ta, tb, d1, d2, i1, i2)
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


GRID_CASES = {
    # what the grid search has to get right: (name -> builder of (a, b))
    "far_apart": lambda r: ((r.normal(size=(1, 3000, 3)) * 2 + 500).astype(np.float32), (r.normal(size=(1, 2500, 3)) * 2 - 500).astype(np.float32)),   # nobody settles within the rings: the brute force behind takes over
    "one_far_query": lambda r: (np.concatenate([r.normal(size=(1, 2000, 3)) * 5, [[[900.0, -700.0, 300.0]]]], 1).astype(np.float32), (r.normal(size=(1, 2500, 3)) * 5).astype(np.float32)),
    "strays": lambda r: (np.concatenate([r.normal(size=(1, 4000, 3)) * 5, r.uniform(-3000, 3000, size=(1, 40, 3))], 1).astype(np.float32),      # 1 % stray returns: searched one by one (round 5), the rest by the grid
                         np.concatenate([r.normal(size=(1, 3500, 3)) * 5, r.uniform(-3000, 3000, size=(1, 25, 3))], 1).astype(np.float32)),
    "flat": lambda r: (np.concatenate([r.uniform(-60, 60, size=(1, 4000, 2)), np.zeros((1, 4000, 1))], 2).astype(np.float32),
                       np.concatenate([r.uniform(-60, 60, size=(1, 3500, 2)), np.zeros((1, 3500, 1))], 2).astype(np.float32)),
    "line": lambda r: (np.concatenate([r.uniform(-80, 80, size=(1, 3000, 1)), np.full((1, 3000, 2), 1.5)], 2).astype(np.float32),
                       np.concatenate([r.uniform(-80, 80, size=(1, 3100, 1)), np.full((1, 3100, 2), 1.5)], 2).astype(np.float32)),
    "clustered": lambda r: ((r.normal(size=(1, 5000, 3)) * np.where(r.random((1, 5000, 1)) < 0.9, 0.05, 40.0)).astype(np.float32),
                            (r.normal(size=(1, 5000, 3)) * np.where(r.random((1, 5000, 1)) < 0.9, 0.05, 40.0)).astype(np.float32)),
    "on_cell_faces": lambda r: ((np.round(r.uniform(-20, 20, size=(1, 4000, 3)) * 2) / 2).astype(np.float32), (np.round(r.uniform(-20, 20, size=(1, 4000, 3)) * 2) / 2).astype(np.float32)),   # lattice points: ties everywhere
    "big_offset": lambda r: ((r.normal(size=(1, 3000, 3)) * 3 + 1.0e4).astype(np.float32), (r.normal(size=(1, 3000, 3)) * 3 + 1.0e4).astype(np.float32)),   # coarse ulps against a small extent
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(GRID_CASES))
def test_hip_grid_search_edge_cases(case, hip_lib_built):
    """The uniform-grid search in front of the brute force (round 4) on inputs built against it; indices and distances bit for bit."""
    a, b = GRID_CASES[case](np.random.default_rng(sum(map(ord, case))))
    ref = chamfer3d.forward(a, b)
    d1, d2, i1, i2 = _hip_forward(a, b)
    assert np.array_equal(i1, ref[2]) and np.array_equal(i2, ref[3]), case
    assert np.array_equal(d1, ref[0]) and np.array_equal(d2, ref[1]), case


@pytest.mark.gpu
def test_hip_grid_search_at_frame_size_against_the_kdtree(hip_lib_built):
    """Two LiDAR-like clouds of one 64 x 2650 frame each (the size PointsMeter feeds it): too many pairs for the numpy oracle, so the
    distances are held against scipy's KD-tree (float64) and the indices by re-evaluating the reference's fp32 expression at them."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(7)
    n = 64 * 2650
    def cloud():
        az = rng.uniform(-np.pi, np.pi, n); el = rng.uniform(-0.31, 0.04, n); r = np.minimum(rng.gamma(2.0, 9.0, n) + 2.0, 80.0)
        return np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1).astype(np.float32)[None]
    a, b = cloud(), cloud()
    b[0, :5000] = a[0, :5000] + rng.normal(scale=0.02, size=(5000, 3)).astype(np.float32)
    d1, d2, i1, i2 = _hip_forward(a, b)
    for q, t, d, i in ((a[0], b[0], d1[0], i1[0]), (b[0], a[0], d2[0], i2[0])):
        assert i.min() >= 0 and i.max() < t.shape[0]
        diff = t[i] - q
        assert np.array_equal(d, (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2])     # the reported neighbour's own fp32 distance
        dd, _ = cKDTree(t.astype(np.float64)).query(q.astype(np.float64))
        assert np.all(d <= (dd ** 2) * (1 + 1e-5) + 1e-9)                                                         # ... and nobody is nearer
