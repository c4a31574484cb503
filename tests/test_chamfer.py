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
