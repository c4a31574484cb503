"""`PointsMeter` -- the reference's point-cloud evaluation metric (/root/reference/utils/lidar_utils.py:234-292) on the device.

Same class, same methods.  `update(preds, truths)` takes the [B, H, W] range images as the reference's does (train.py:367) and runs
the whole metric -- back-projection of the non-empty pixels, nearest neighbours both ways, means, F-score at 0.05 -- in ONE native call
on device-resident images (include/lidargs_chamfer.h lidargs_points_meter); the reference copies both images to the host, back-projects
with numpy, uploads the clouds for its chamfer kernel and reads the results back (:256-279).  The per-frame values stay on the device
until `measure()` asks for them."""
import ctypes as C
import os

import numpy as np
import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
_lib.lidargs_points_meter.restype = C.c_int
_lib.lidargs_points_meter_scratch_bytes.restype = C.c_size_t


def points_metrics(pred, truth, scale=1.0, intrinsics=None, beam_inclinations=None, threshold=0.05):
    """pred, truth: [H, W] float32 range images on the device.  -> float32[6] on the device:
    (chamfer distance = dist1.mean() + dist2.mean(), F-score, precision, recall, points of pred, points of truth)."""
    _base._require_device(pred, "preds"); _base._require_device(truth, "truths")
    if pred.shape != truth.shape or pred.ndim != 2:
        raise RuntimeError("points_metrics: preds and truths must be [H, W] images of one shape")
    dev = pred.device
    p32 = lambda t: t.detach().to(torch.float32).contiguous()
    pred, truth = p32(pred), p32(truth)
    H, W = int(pred.shape[0]), int(pred.shape[1])
    beams = None
    fov_up = fov = 0.0
    if beam_inclinations is not None:
        beams = torch.as_tensor(np.asarray(beam_inclinations.detach().cpu() if torch.is_tensor(beam_inclinations) else beam_inclinations),
                                dtype=torch.float32).to(dev).contiguous() if not (torch.is_tensor(beam_inclinations) and beam_inclinations.is_cuda) \
            else beam_inclinations.detach().to(torch.float32).contiguous()
        if beams.numel() != H:
            raise RuntimeError("points_metrics: beam_inclinations must have one entry per image row")
    elif intrinsics is not None:
        fov_up, fov = float(intrinsics[0]), float(intrinsics[1])
    else:
        raise RuntimeError("points_metrics: need beam_inclinations or intrinsics = (fov_up, fov)")
    out = torch.empty(6, dtype=torch.float32, device=dev)
    nb = int(_lib.lidargs_points_meter_scratch_bytes(C.c_int(H), C.c_int(W)))
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lidargs_points_meter(C.c_int(H), C.c_int(W), _base._ptr(pred), _base._ptr(truth), C.c_float(float(scale)), _base._ptr(beams),
                                       C.c_float(fov_up), C.c_float(fov), C.c_float(float(threshold)), _base._ptr(out), _base._ptr(scratch),
                                       C.c_size_t(nb), _base._stream(dev))
    if rc < 0:
        _base._raise(rc, "points_meter")
    return out


class PointsMeter:
    """utils/lidar_utils.py:234-292, same interface; V holds device tensors until measure()."""

    def __init__(self, scale, intrinsics, beam_inclinations=None):
        self.V = []
        self.N = 0
        self.scale = scale
        self.intrinsics = intrinsics
        self.beam_inclinations = beam_inclinations

    def clear(self):
        self.V = []
        self.N = 0

    def update(self, preds, truths):
        # the reference takes device tensors, host tensors or numpy arrays alike (prepare_inputs, :247-254) and ends on the device either
        # way (:268-270): host inputs are moved there.  Beam tables are evaluated in float32 (a float64 table is rounded first; the
        # reference keeps numpy's float64 until torch.FloatTensor, :269 -- a difference of at most half a float32 ulp of the angle,
        # ~3e-8 relative in the points, far inside the metric's seven pinned digits).
        dev = next((t.device for t in (preds, truths) if torch.is_tensor(t) and t.is_cuda), torch.device("cuda"))
        preds, truths = (t.to(dev) if torch.is_tensor(t) else torch.as_tensor(np.asarray(t)).to(dev) for t in (preds, truths))
        out = points_metrics(preds[0], truths[0], self.scale, self.intrinsics, self.beam_inclinations, threshold=0.05)   # [B, H, W]: image 0, as the reference
        self.V.append(out[:2])                                         # (chamfer_dis, f_score), :280
        self.N += 1

    def measure(self):
        assert self.N == len(self.V)
        return torch.stack(self.V).mean(0).cpu().numpy() if self.V else np.array([np.nan, np.nan])

    def write(self, writer, global_step, prefix=""):
        writer.add_scalar(os.path.join(prefix, "CD"), self.measure()[0], global_step)

    def report(self):
        return f'CD f-score = {self.measure()}'
