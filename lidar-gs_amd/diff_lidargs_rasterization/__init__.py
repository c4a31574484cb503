"""diff_lidargs_rasterization -- MI355X-native drop-in for the reference package of the same name.

Public surface (what gaussian_renderer/__init__.py:121 imports and :145-179, :252-257 call):

    GaussianRasterizationSettings   15-field NamedTuple, same field names and order as
                                    R3/diff_lidargs_rasterization/__init__.py:164-179
    GaussianRasterizer              nn.Module with forward / visible_filter / markVisible (:182-263)
    rasterize_gaussians             functional form (:21-42)

`forward` returns (color[2,H,W], depth[1,H,W], occ[1,H,W], radii[P]) and is differentiable
w.r.t. means3D, means2D (a gradient sink: .grad[:, :2] / [:, 2] feed densification,
scene/gaussian_model.py:617-619), colors_precomp, opacities, scales, rotations, cov3D_precomp.
The native side is lidar-gs_amd/csrc/*.hip behind the C ABI of include/lidargs_rasterizer.h.
"""
from typing import NamedTuple

import os

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor        # transposed world->lidar 4x4 (scene/cameras.py:56)
    projmatrix: torch.Tensor        # accepted, unused by the LiDAR path
    sh_degree: int
    campos: torch.Tensor            # accepted, unused by the LiDAR path
    prefiltered: bool
    beam_inclinations: torch.Tensor  # [H] ascending radians
    lidar_far: int
    lidar_near: int
    debug: bool


def _snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


class _EnqueueState:
    """Caller-side state of the enqueue-only forward (lidargs_forward_enqueue): the binning capacity and tile height, learnt from
    one ordinary frame and then kept with headroom, plus the pinned status words of the last enqueue-only frame.  It lives in the
    GaussianRasterizer module (the caller), not in the library."""
    HEADROOM = 1.25

    def __init__(self):
        self.cap, self.th = None, 4
        self.status = None            # pinned int32[16]
        self.event, self.pending = None, False
        self.frames = 0

    def learn(self, num_rendered):
        """From an ordinary frame's num_rendered (instance capacity | tile-height code)."""
        need = int(num_rendered) & ~3
        self.th = 4 << (int(num_rendered) & 3)
        self.cap = max((int(need * self.HEADROOM) + 4096 + 3) & ~3, self.cap or 0)      # multiples of 4: what num_rendered can carry

    def plan(self):
        """(capacity, tile_rows, status) for the next frame, or None while nothing has been learnt.  Looks at the status of
        the previous enqueue-only frame if the stream has passed it (never waits): grows the capacity early, and raises if
        that frame overflowed -- its outputs were wrong."""
        if self.cap is None:
            return None
        capturing = torch.cuda.is_current_stream_capturing()
        if self.pending and not capturing and self.event.query():
            self.pending = False
            need, over = int(self.status[0]), int(self.status[8])
            if need * 1.08 > self.cap:
                self.cap = (int(need * self.HEADROOM) + 4096 + 3) & ~3
            if over:
                raise RuntimeError(f"diff_lidargs_rasterization: an enqueue-only frame needed {need} list instances but its binning "
                                   f"buffer held {int(self.status[9])}; that frame's outputs are invalid (capacity raised to {self.cap}, "
                                   "re-render it)")
        if self.status is None:
            self.status = torch.zeros(16, dtype=torch.int32).pin_memory()
        return self.cap, self.th, self.status

    def submitted(self):
        if torch.cuda.is_current_stream_capturing():
            return
        if self.event is None:
            self.event = torch.cuda.Event()
        self.event.record()
        self.pending = True
        self.frames += 1

    def read_status(self):
        """Synchronises; dict of the last enqueue-only frame's status words."""
        torch.cuda.synchronize()
        st = self.status.tolist() if self.status is not None else [0] * 16
        return dict(needed=st[0], binned=st[1], overflow=bool(st[8]), capacity=st[9], tile_rows=self.th)


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd node: marshals to `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` with the
    argument order of R3/diff_lidargs_rasterization/__init__.py:60-81 and :113-136."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, enqueue_state=None):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.image_height, rs.image_width, rs.beam_inclinations, sh, rs.sh_degree,
                rs.campos, rs.prefiltered, rs.lidar_far, rs.lidar_near, rs.debug)
        plan = enqueue_state.plan() if enqueue_state is not None else None     # None: an ordinary frame (one host wait)
        if rs.debug:
            saved = _snapshot(args)  # copied before anything can corrupt them
            try:
                out = _C.rasterize_gaussians(*args, enqueue=plan)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _C.rasterize_gaussians(*args, enqueue=plan)
        num_rendered, color, depth, occ, radii, geom_buffer, binning_buffer, img_buffer = out
        if enqueue_state is not None:
            if plan is None:
                enqueue_state.learn(num_rendered)
            else:
                enqueue_state.submitted()
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geom_buffer, binning_buffer, img_buffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)      # an output nobody used arrives as None in backward (handled there), not as a zero plane
        # no zero tensors made for outputs the loss does not use (radii never has a gradient: that alone is one fill of P ints a
        # frame); backward fills in the image planes that are missing
        ctx.set_materialize_grads(False)
        return color, depth, occ, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, grad_out_occ, _grad_radii):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buffer, binning_buffer, img_buffer = ctx.saved_tensors
        plane = lambda g, c: g if g is not None else torch.zeros((c, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        grad_out_color, grad_out_depth, grad_out_occ = plane(grad_out_color, 2), plane(grad_out_depth, 1), plane(grad_out_occ, 1)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.beam_inclinations, rs.tanfovx, rs.tanfovy, grad_out_color, grad_out_depth, grad_out_occ,
                sh, rs.sh_degree, rs.campos, geom_buffer, ctx.num_rendered, binning_buffer, img_buffer, rs.debug)
        # dL/dcov3D is an output only when the covariance was an input; with scales + rotations nobody receives it
        want_cov = cov3Ds_precomp.numel() != 0
        if rs.debug:
            saved = _snapshot(args)
            try:
                grads = _C.rasterize_gaussians_backward(*args, want_cov3D_grad=want_cov)
            except Exception:
                torch.save(saved, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            grads = _C.rasterize_gaussians_backward(*args, want_cov3D_grad=want_cov)
        grad_means2D, grad_colors, grad_opacities, grad_means3D, grad_cov3Ds, grad_sh, grad_scales, grad_rotations = grads
        # one slot per forward input, in forward's order (R3/.../__init__.py:150-160)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        enqueue_state=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, enqueue_state)


def _or_empty(t):
    # the reference substitutes CPU empties, `torch.Tensor([])` (R3/.../__init__.py:208-218); an empty tensor
    # becomes a NULL pointer at the C boundary whatever its device
    return torch.Tensor([]) if t is None else t


class GaussianRasterizer(nn.Module):
    """Same constructor and methods as the reference's module (R3/diff_lidargs_rasterization/__init__.py:181-263).

    One addition, off by default: `enqueue_only` (attribute, or LIDARGS_ENQUEUE_ONLY=1 in the environment).  The reference's
    forward blocks the host once per frame (cudaMemcpy of the instance count, R3/cr/rasterizer_impl.cu:292) and so does the
    default forward here; with enqueue_only the first frame runs that way and tells the module how many list instances the
    view needs, every later frame only ENQUEUES work (lidargs_forward_enqueue: capacity = 1.25 x what was needed, counts stay
    on the device) -- the host runs ahead and the frame can be captured in a HIP graph.  The price: a view that suddenly needs
    more than the capacity is detected one frame late (RuntimeError from the next forward; `enqueue_status()` asks now)."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings
        self.enqueue_only = os.environ.get("LIDARGS_ENQUEUE_ONLY", "0") == "1"
        self._enqueue = _EnqueueState()

    def enqueue_status(self):
        """Synchronises the device; {needed, binned, overflow, capacity, tile_rows} of the last enqueue-only frame."""
        return self._enqueue.read_status()

    def markVisible(self, positions):
        """bool[P]: view-space z > 0.2, the camera-style test the reference keeps (R3/cr/auxiliary.h:175-200)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        return rasterize_gaussians(means3D, means2D, _or_empty(shs), _or_empty(colors_precomp), opacities,
                                   _or_empty(scales), _or_empty(rotations), _or_empty(cov3D_precomp), self.raster_settings,
                                   self._enqueue if self.enqueue_only else None)

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        """radii[P] int32 of the cull/footprint test only (prefilter_voxel, gaussian_renderer/__init__.py:252-257)."""
        rs = self.raster_settings
        with torch.no_grad():
            return _C.rasterize_aussians_filter(
                means3D, _or_empty(scales), _or_empty(rotations), rs.scale_modifier, _or_empty(cov3D_precomp),
                rs.viewmatrix, rs.projmatrix, rs.campos, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
                rs.beam_inclinations, rs.prefiltered, rs.lidar_far, rs.lidar_near, rs.debug)
