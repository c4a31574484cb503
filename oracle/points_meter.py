"""CPU restatement of PointsMeter.update (/root/reference/utils/lidar_utils.py:253-282).  TEST INFRASTRUCTURE ONLY.

pano -> points is oracle/range_view.py (pinned: tests/golden/rangeview_golden.npz), the nearest neighbours oracle/chamfer3d.py (the
kernel's definition; the CUDA extension cannot run here), the means and the F-score (extern/fscore.py:4-18) float32 as torch evaluates
them.  Pinned as a whole by tests/golden/points_meter_golden.npz (tests/golden/make_points_meter_golden.py executes the reference's
pano_to_lidar and fscore)."""
import numpy as np

from . import chamfer3d, range_view


def update(pred, truth, beams, scale=1.0, threshold=0.05):
    """-> (chamfer_dis, f_score, precision, recall, n, m) for one pair of [H, W] range images."""
    pred = (np.asarray(pred, np.float32) / np.float32(scale)).astype(np.float32)
    truth = (np.asarray(truth, np.float32) / np.float32(scale)).astype(np.float32)
    beams = np.asarray(beams, np.float32)
    p1 = range_view.pano_to_points(pred, np.zeros_like(pred), beams)[:, :3].astype(np.float32)
    p2 = range_view.pano_to_points(truth, np.zeros_like(truth), beams)[:, :3].astype(np.float32)
    d1, d2, _, _ = chamfer3d.forward(p1[None], p2[None])
    cd = np.float32(d1.mean(dtype=np.float32)) + np.float32(d2.mean(dtype=np.float32))
    pr1 = np.float32((d1 < threshold).astype(np.float32).mean()); pr2 = np.float32((d2 < threshold).astype(np.float32).mean())
    f = np.float32(2) * pr1 * pr2 / (pr1 + pr2) if (pr1 + pr2) > 0 else np.float32(0)
    return float(cd), float(f), float(pr1), float(pr2), int(p1.shape[0]), int(p2.shape[0])
