"""PointsMeter (second half; /root/reference/utils/lidar_utils.py:234-292): the oracle against the fixture
made by executing the reference's pano_to_lidar and fscore (tests/golden/make_points_meter_golden.py), and the native one-call metric
(lidar-gs_amd/points_meter.py -> lidargs_points_meter) against both.

Tolerance (floating point, stated): the clouds' points agree to 1e-6 relative (the pixel rays' cos / sin are correctly rounded on the
device, numpy's float32 cos / sin are within an ulp of that), so every squared distance agrees to ~1e-5 and the chamfer distance --
a mean of them -- is held to 2e-5 relative; precision / recall are counts of `dist < 0.05` over n points: a point within 1e-5 of the
threshold may fall on either side, so they are held to 2 / n."""
import os

import numpy as np
import pytest

from oracle import points_meter as pm_oracle
from oracle import range_view

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "points_meter_golden.npz"))
TAGS = ["a", "b", "c"]        # (tag d uses the (fov_up, fov) intrinsics instead of a beam table)


def _case(tag):
    H, W = int(GOLD[f"{tag}_H"]), int(GOLD[f"{tag}_W"])
    beams = GOLD[f"{tag}_beams"]
    return H, W, GOLD[f"{tag}_pred"], GOLD[f"{tag}_truth"], (beams if beams.size else None), float(GOLD[f"{tag}_scale"]), tuple(GOLD[f"{tag}_fov"])


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_the_reference_fixture(tag):
    H, W, pred, truth, beams, scale, _ = _case(tag)
    p = (pred / np.float32(scale)).astype(np.float32)
    pts = range_view.pano_to_points(p, np.zeros_like(p), beams)[:, :3].astype(np.float32)
    np.testing.assert_array_equal(pts, GOLD[f"{tag}_pred_lidar"])       # the back-projection, bit for bit (the same numpy)
    cd, f, pr, rc, n, m = pm_oracle.update(pred, truth, beams, scale)
    assert (n, m) == (GOLD[f"{tag}_pred_lidar"].shape[0], GOLD[f"{tag}_gt_lidar"].shape[0])
    assert abs(cd - float(GOLD[f"{tag}_cd"])) <= 2e-6 * abs(cd)          # numpy's against torch's float32 mean
    assert abs(f - float(GOLD[f"{tag}_fscore"])) <= 1e-6 and abs(pr - float(GOLD[f"{tag}_precision"])) <= 1e-6 and abs(rc - float(GOLD[f"{tag}_recall"])) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS + ["d"])
def test_native_points_meter_matches_the_reference_fixture(tag, hip_lib_built):
    import torch
    import points_meter
    H, W, pred, truth, beams, scale, fov = _case(tag)
    out = points_meter.points_metrics(torch.from_numpy(pred).cuda(), torch.from_numpy(truth).cuda(), scale=scale,
                                      intrinsics=None if beams is not None else fov, beam_inclinations=beams).cpu().numpy()
    n, m = GOLD[f"{tag}_pred_lidar"].shape[0], GOLD[f"{tag}_gt_lidar"].shape[0]
    print(tag, out, float(GOLD[f"{tag}_cd"]), float(GOLD[f"{tag}_fscore"]))
    assert (int(out[4]), int(out[5])) == (n, m)
    assert abs(out[0] - float(GOLD[f"{tag}_cd"])) <= 2e-5 * float(GOLD[f"{tag}_cd"])
    assert abs(out[2] - float(GOLD[f"{tag}_precision"])) <= 2.0 / n and abs(out[3] - float(GOLD[f"{tag}_recall"])) <= 2.0 / m
    assert abs(out[1] - float(GOLD[f"{tag}_fscore"])) <= 2.0 / min(n, m)


@pytest.mark.gpu
def test_points_meter_class_and_edge_cases(hip_lib_built):
    """The class as train.py:354-372 uses it (update per frame, measure at the end), an all-empty prediction (the reference's NaN mean,
    F-score 0) and images without a single empty pixel."""
    import torch
    import points_meter
    H, W, pred, truth, beams, scale, _ = _case("a")
    meter = points_meter.PointsMeter(scale=scale, intrinsics=None, beam_inclinations=torch.from_numpy(beams).cuda())
    meter.update(torch.from_numpy(pred).cuda()[None], torch.from_numpy(truth).cuda()[None])
    meter.update(torch.from_numpy(truth).cuda()[None], torch.from_numpy(truth).cuda()[None])       # identical clouds: distance 0, F-score 1
    v = meter.measure()
    assert abs(v[0] - 0.5 * float(GOLD["a_cd"])) <= 2e-5 * float(GOLD["a_cd"]) and abs(v[1] - 0.5 * (float(GOLD["a_fscore"]) + 1.0)) <= 2e-3
    assert "CD f-score" in meter.report()
    # host tensors and numpy arrays, float64 beam table: accepted like the reference's prepare_inputs does (:247-254), same numbers
    host = points_meter.PointsMeter(scale=scale, intrinsics=None, beam_inclinations=beams.astype(np.float64))
    host.update(torch.from_numpy(pred)[None], truth[None])
    host.update(truth[None], torch.from_numpy(truth).cuda()[None])
    assert np.allclose(host.measure(), v, rtol=1e-6, atol=1e-7)
    empty = points_meter.points_metrics(torch.zeros(H, W).cuda(), torch.from_numpy(truth).cuda(), beam_inclinations=beams).cpu().numpy()
    assert np.isnan(empty[0]) and empty[1] == 0.0 and empty[4] == 0
    full = np.abs(truth) + 1.0
    o = points_meter.points_metrics(torch.from_numpy(full).cuda(), torch.from_numpy(full).cuda(), beam_inclinations=beams).cpu().numpy()
    assert o[0] == 0.0 and o[1] == 1.0 and int(o[4]) == H * W


@pytest.mark.gpu
def test_points_meter_at_frame_size_against_the_step_by_step_path(hip_lib_built):
    """One 64 x 2650 frame (the size train.py feeds it; too many pairs for the numpy oracle): the one-call metric against the same steps
    taken one by one on the device -- torch's compaction of the non-empty pixels, chamfer_3D.forward (held bit for bit against the oracle
    and a KD-tree in tests/test_chamfer.py), torch's means."""
    import torch
    import chamfer_3D
    import points_meter
    import lidargs_scenes as sc
    H, W = 64, 2650
    rng = np.random.default_rng(5)
    beams = np.ascontiguousarray(sc.beam_table(H, "waymo"), dtype=np.float32)
    truth = (rng.gamma(2.0, 9.0, size=(H, W)) + 2.0).astype(np.float32); truth[rng.random((H, W)) < 0.2] = 0.0
    pred = (truth * (1.0 + 0.005 * rng.normal(size=(H, W)))).astype(np.float32); pred[rng.random((H, W)) < 0.1] = 0.0
    out = points_meter.points_metrics(torch.from_numpy(pred).cuda(), torch.from_numpy(truth).cuda(), beam_inclinations=beams).cpu().numpy()
    clouds = []
    for img in (pred, truth):
        pts = range_view.pano_to_points(img, np.zeros_like(img), beams)[:, :3].astype(np.float32)
        clouds.append(torch.from_numpy(pts).cuda()[None].contiguous())
    n, m = clouds[0].shape[1], clouds[1].shape[1]
    d1, d2 = torch.empty(1, n, device="cuda"), torch.empty(1, m, device="cuda")
    i1, i2 = torch.empty(1, n, dtype=torch.int32, device="cuda"), torch.empty(1, m, dtype=torch.int32, device="cuda")
    chamfer_3D.forward(clouds[0], clouds[1], d1, d2, i1, i2)
    cd = float(d1.mean() + d2.mean()); p1 = float((d1 < 0.05).float().mean()); p2 = float((d2 < 0.05).float().mean())
    assert (int(out[4]), int(out[5])) == (n, m)
    assert abs(out[0] - cd) <= 2e-5 * cd and abs(out[2] - p1) <= 3.0 / n and abs(out[3] - p2) <= 3.0 / m
    assert abs(out[1] - 2 * p1 * p2 / (p1 + p2)) <= 1e-4
