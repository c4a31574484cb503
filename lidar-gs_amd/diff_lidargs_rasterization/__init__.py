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


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd node: marshals to `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` with the
    argument order of R3/diff_lidargs_rasterization/__init__.py:60-81 and :113-136."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.image_height, rs.image_width, rs.beam_inclinations, sh, rs.sh_degree,
                rs.campos, rs.prefiltered, rs.lidar_far, rs.lidar_near, rs.debug)
        if rs.debug:
            saved = _snapshot(args)  # copied before anything can corrupt them
            try:
                out = _C.rasterize_gaussians(*args)
            except Exception:
                torch.save(saved, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _C.rasterize_gaussians(*args)
        num_rendered, color, depth, occ, radii, geom_buffer, binning_buffer, img_buffer = out
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              geom_buffer, binning_buffer, img_buffer)
        ctx.mark_non_differentiable(radii)
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
        if rs.debug:
            saved = _snapshot(args)
            try:
                grads = _C.rasterize_gaussians_backward(*args)
            except Exception:
                torch.save(saved, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            grads = _C.rasterize_gaussians_backward(*args)
        grad_means2D, grad_colors, grad_opacities, grad_means3D, grad_cov3Ds, grad_sh, grad_scales, grad_rotations = grads
        # one slot per forward input, in forward's order (R3/.../__init__.py:150-160)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


def _or_empty(t):
    # the reference substitutes CPU empties, `torch.Tensor([])` (R3/.../__init__.py:208-218); an empty tensor
    # becomes a NULL pointer at the C boundary whatever its device
    return torch.Tensor([]) if t is None else t


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

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
                                   _or_empty(scales), _or_empty(rotations), _or_empty(cov3D_precomp), self.raster_settings)

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        """radii[P] int32 of the cull/footprint test only (prefilter_voxel, gaussian_renderer/__init__.py:252-257)."""
        rs = self.raster_settings
        with torch.no_grad():
            return _C.rasterize_aussians_filter(
                means3D, _or_empty(scales), _or_empty(rotations), rs.scale_modifier, _or_empty(cov3D_precomp),
                rs.viewmatrix, rs.projmatrix, rs.campos, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width,
                rs.beam_inclinations, rs.prefiltered, rs.lidar_far, rs.lidar_near, rs.debug)
