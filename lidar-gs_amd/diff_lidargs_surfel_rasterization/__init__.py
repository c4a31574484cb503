"""Drop-in for the reference's `diff_lidargs_surfel_rasterization` package (BASELINE config 5).

Same public surface as R2/diff_lidargs_surfel_rasterization/__init__.py: `GaussianRasterizationSettings`
(14 fields, :188-202), `GaussianRasterizer` with forward / markVisible / visible_filter (:204-273) and the
functional `rasterize_gaussians` (:21-43).  forward returns (color[2,H,W], radii[P], others[7,H,W], pixels[P,1]);
gradients flow to means3D, means2D (the [P,4] densification statistics), colors_precomp, opacities, scales[P,2]
and rotations.  A precomputed transform (`cov3Ds_precomp` = transMat_precomp [P,9]) is honoured as the reference honours it --
given NEXT TO scales / rotations through the functional `rasterize_gaussians`, it replaces the rows the blends use, and receives
dL_dtransMat; spherical harmonics are not part of the LiDAR path (rejected natively).
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    bg: torch.Tensor
    scale_modifier: float
    depth_threshold: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    beam_inclinations: torch.Tensor
    lidar_far: int
    lidar_near: int
    debug: bool


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        num_rendered, color, others, radii, pixels, geom, binning, img = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.beam_inclinations, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered,
            rs.lidar_far, rs.lidar_near, rs.debug)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii, pixels)
        ctx.set_materialize_grads(False)      # no zero tensors for radii / pixels; backward fills in a missing image gradient
        return color, radii, others, pixels

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_others, grad_pix):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        plane = lambda g, c: g if g is not None else torch.zeros((c, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        grad_out_color, grad_others = plane(grad_out_color, 2), plane(grad_others, 7)
        (grad_means2D, grad_colors, grad_opacities, grad_means3D, grad_transMat, grad_sh, grad_scales, grad_rotations,
         _depth) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
            rs.projmatrix, rs.beam_inclinations, grad_out_color, grad_others, sh, rs.sh_degree, rs.campos, geom,
            ctx.num_rendered, binning, img, rs.debug, want_intermediates=bool(cov3Ds_precomp.numel()))
        return (grad_means3D, grad_means2D, grad_sh if sh.numel() else None, grad_colors, grad_opacities, grad_scales,
                grad_rotations, grad_transMat if cov3Ds_precomp.numel() else None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
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
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        return rasterize_gaussians(
            means3D, means2D, empty if shs is None else shs, empty if colors_precomp is None else colors_precomp, opacities,
            empty if scales is None else scales, empty if rotations is None else rotations,
            empty if cov3D_precomp is None else cov3D_precomp, self.raster_settings)

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        with torch.no_grad():
            return _C.rasterize_aussians_filter(
                means3D, empty if scales is None else scales, empty if rotations is None else rotations, rs.scale_modifier,
                empty if cov3D_precomp is None else cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.beam_inclinations,
                rs.image_height, rs.image_width, rs.prefiltered, rs.lidar_far, rs.lidar_near, rs.debug)
