"""Restatement of the reference's numpy range-view projector (TEST INFRASTRUCTURE ONLY).

This is the reference's only Python/CPU path over the range-view geometry
(utils/lidar_utils.py:33-110, :171-232, :296-299 == R3/python_imp/imp.py:28-131): a per-point
Python loop, single core by construction.  It is what BASELINE.md calls baseline B2 and what
the north star means by "the reference's Python/CPU preprocess path".  Pinned against
tests/golden/rangeview_golden.npz, which was produced by executing the reference functions.

Conventions it fixes (and that the rasterizer shares, R3/cr/forward.cu:333-359, :589-591):
  column  c = (pi - atan2(y, x)) / (2 pi / W)          azimuth pi at column 0, decreasing
  row     from the beam whose inclination is closest, rows counted from the TOP beam
  ray     of pixel (i, j): elevation beams[::-1][j], azimuth -(i - W/2)/W * 2 pi
"""
from bisect import bisect_left

import numpy as np


def nearest_beam(beams, angle):
    """utils/lidar_utils.py:33-49 -- NEAREST beam (the CUDA kernel bisects left instead,
    R3/cr/auxiliary.h:41-63)."""
    n = len(beams)
    if angle >= beams[n - 1]:
        return n - 1
    if angle <= beams[0]:
        return 0
    hi = bisect_left(beams, angle)
    lo = hi - 1
    return hi if (beams[hi] - angle) < (angle - beams[lo]) else lo


def points_to_pano(points_xyzi, H, W, beams, max_depth=80):
    """utils/lidar_utils.py:51-110 with beam_inclinations given: z-buffer (min range) projection.

    Note the reference writes row H - beam_index (so beam 0 falls off the image) while
    pano_to_points reads row H-1-beam_index; that one-row offset is the reference's behaviour."""
    xyz = points_xyzi[:, :3]
    inten = points_xyzi[:, 3]
    ranges = np.linalg.norm(xyz, axis=1)
    pano = np.zeros((H, W))
    out_i = np.zeros((H, W))
    col_step = 2 * np.pi / W
    for p, rng_, val in zip(xyz, ranges, inten):
        if rng_ >= max_depth:
            continue
        px, py, pz = p
        col = int(round((np.pi - np.arctan2(py, px)) / col_step))
        elev = np.arctan2(pz, np.sqrt(px ** 2 + py ** 2))
        row = H - nearest_beam(beams, elev)
        if row >= H or row < 0 or col >= W or col < 0:
            continue
        cur = pano[row, col]
        if cur == 0.0 or cur > rng_:
            pano[row, col] = rng_
            out_i[row, col] = val
    return pano, out_i


def pixel_rays(H, W, beams):
    """Unit ray of every pixel, [H,W,3] (utils/lidar_utils.py:186-199)."""
    i, j = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    az = -(i - W / 2.0) / W * 2.0 * np.pi
    el = np.expand_dims(np.asarray(beams)[::-1], 1).repeat(W, 1)
    return np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], -1)


def pano_to_points(pano, intensities, beams):
    """utils/lidar_utils.py:171-214: back-project non-empty pixels -> [n,4] (x,y,z,intensity)."""
    H, W = pano.shape
    pts = pixel_rays(H, W, beams) * pano.reshape(H, W, 1)
    full = np.concatenate([pts, intensities.reshape(H, W, 1)], axis=2)
    return full[np.where(pano != 0.0)]


def fov_beam_table(fov_up, fov, H):
    """utils/lidar_utils.py:296-299 get_beam_inclinations."""
    j = np.arange(H, dtype=np.float32)
    return ((fov_up - j / H * fov) / 180 * np.pi)[::-1]
