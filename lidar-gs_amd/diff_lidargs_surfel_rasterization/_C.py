"""`diff_lidargs_surfel_rasterization._C` -- native module of the surfel (2DGS laser-surfel) drop-in package.

Binds the four functions the reference's extension exports (R2/ext.cpp: rasterize_gaussians,
rasterize_gaussians_backward, mark_visible, rasterize_aussians_filter; R2 = /root/reference/submodules/
diff_lidargs_surfel_rasterization) with the same names, positional signatures and return tuples
(R2/rasterize_points.cu:46-141, :143-242, :244-263, :265-330) to lidargs_surfel_* of
include/lidargs_rasterizer.h.  The shared library, allocator callback and helpers are those of
diff_lidargs_rasterization._C; there is no CPU path.
"""
import ctypes as C

import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
_alloc_cb, _Scratch, _f32, _ptr, _stream, _require_device, _raise = (
    _base._alloc_cb, _base._Scratch, _base._f32, _base._ptr, _base._stream, _base._require_device, _base._raise)
NUM_CHANNELS = 2   # R2/cr/config.h
for _name in ("lidargs_surfel_forward", "lidargs_surfel_backward", "lidargs_surfel_visible_filter"):
    getattr(_lib, _name).restype = C.c_int


def _dev_or_none(t):
    return _ptr(t if (t is not None and t.is_cuda) else None)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, transMat_precomp,
                        viewmatrix, projmatrix, beam_inclinations, image_height, image_width, sh, degree, campos,
                        prefiltered, lidar_far, lidar_near, debug):
    """RasterizeGaussiansCUDA (R2/rasterize_points.cu:46-141).

    Returns (num_rendered, out_color[2,H,W], out_others[7,H,W], radii[P] int32, pixels[P,1],
    geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_device(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    mk = torch.empty if P != 0 else torch.zeros                       # the library writes every pixel and every row (as the 3-D binding relies on)
    out_color = mk((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_others = mk((7, H, W), dtype=torch.float32, device=dev)
    radii = mk((P,), dtype=torch.int32, device=dev)
    radii_xy = mk((2 * P,), dtype=torch.int32, device=dev)
    pixels = torch.zeros((P, 1), dtype=torch.float32, device=dev)      # never written, by the reference either (R2/cr/forward.cu:522): zeros
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    rendered = 0
    if P != 0:
        M = int(sh.size(1)) if sh.numel() != 0 and sh.ndim > 1 else 0
        bg, m3, col, opa = _f32(background, "background"), _f32(means3D, "means3D"), _f32(colors, "colors"), _f32(opacity, "opacity")
        sc, rot, tm = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(transMat_precomp, "transMat_precomp")
        vm, beams = _f32(viewmatrix, "viewmatrix"), _f32(beam_inclinations, "beam_inclinations")
        for t, n in ((bg, "bg"), (col, "colors_precomp"), (opa, "opacities"), (vm, "viewmatrix"), (beams, "beam_inclinations")):
            if t.numel():
                _require_device(t, n)
        with torch.cuda.device(dev):
            rendered = _lib.lidargs_surfel_forward(
                _alloc_cb, geom.user, _alloc_cb, binning.user, _alloc_cb, img.user,
                C.c_int(P), C.c_int(int(degree)), C.c_int(M), _ptr(bg), C.c_int(W), C.c_int(H),
                _ptr(m3), None, _ptr(col), _ptr(opa), _dev_or_none(sc), C.c_float(float(scale_modifier)), _dev_or_none(rot),
                _dev_or_none(tm), _ptr(vm), None, None, _ptr(beams),
                C.c_int(int(bool(prefiltered))), C.c_int(int(lidar_far)), C.c_int(int(lidar_near)),
                _ptr(out_color), _ptr(out_others), _ptr(pixels), _ptr(radii), _ptr(radii_xy),
                C.c_int(int(bool(debug))), _stream(dev))
        if rendered < 0:
            geom.take(); binning.take(); img.take()
            _raise(rendered, "rasterize_gaussians (surfel)")
    return rendered, out_color, out_others, radii, pixels, geom.take(), binning.take(), img.take()


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, transMat_precomp,
                                 viewmatrix, projmatrix, beam_inclinations, dL_dout_color, dL_dout_others, sh, degree,
                                 campos, geomBuffer, R, binningBuffer, imageBuffer, debug, want_intermediates=True):
    """RasterizeGaussiansBackwardCUDA (R2/rasterize_points.cu:143-242).

    Returns (dL_dmeans2D[P,4], dL_dcolors[P,2], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dtransMat[P,9],
    dL_dsh[P,M,3], dL_dscales[P,2], dL_drotations[P,4], depth[P,1]).  want_intermediates=False: dL_dtransMat and the two
    scratch gradients of the reference's kernel pair are not materialised (None is returned for dL_dtransMat)."""
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 and sh.ndim > 1 else 0
    skip = not want_intermediates
    widths = (3, 4, NUM_CHANNELS, 0 if skip else 3, 1, 0 if skip else 9, 0 if skip else 3, 2, 4, 1)   # the reference's torch::zeros list (:193-203), one slab here
    slab = torch.empty(P * sum(widths), dtype=torch.float32, device=dev)   # every row is written by the library
    parts, o = [], 0
    for w in widths:
        parts.append(slab[o:o + P * w].view(P, w)); o += P * w
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dnormal, dL_dopacity, dL_dtransMat, dL_dtm2d, dL_dscales, dL_drotations, depth = parts
    dL_dsh = torch.zeros((P, M, 3), dtype=torch.float32, device=dev)
    if P != 0:
        bg, m3 = _f32(background, "background"), _f32(means3D, "means3D")
        tm, vm, beams = _f32(transMat_precomp, "transMat_precomp"), _f32(viewmatrix, "viewmatrix"), _f32(beam_inclinations, "beam_inclinations")
        g0, g1 = _f32(dL_dout_color, "dL_dout_color"), _f32(dL_dout_others, "dL_dout_others")
        sc, rot, rad = _f32(scales, "scales"), _f32(rotations, "rotations"), radii.contiguous()
        with torch.cuda.device(dev):
            rc = _lib.lidargs_surfel_backward(
                C.c_int(P), C.c_int(int(degree)), C.c_int(M), C.c_int(int(R)), _ptr(bg), C.c_int(W), C.c_int(H),
                _ptr(m3), None, None, _dev_or_none(sc), C.c_float(float(scale_modifier)), _dev_or_none(rot), _dev_or_none(tm),
                _ptr(vm), None, None, _ptr(beams), _ptr(rad), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer),
                _ptr(g0), _ptr(g1), _ptr(dL_dmeans2D), _ptr(dL_dnormal), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_dmeans3D),
                _ptr(dL_dtransMat), _ptr(dL_dtm2d), None, _ptr(dL_dscales), _ptr(dL_drotations), _ptr(depth),
                C.c_int(int(bool(debug))), _stream(dev))
        if rc < 0:
            _raise(rc, "rasterize_gaussians_backward (surfel)")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, (None if skip else dL_dtransMat), dL_dsh, dL_dscales, dL_drotations, depth


def rasterize_aussians_filter(means3D, scales, rotations, scale_modifier, transMat_precomp, viewmatrix, projmatrix,
                              beam_inclinations, image_height, image_width, prefiltered, lidar_far, lidar_near, debug):
    """RasterizeGaussiansfilterCUDA (R2/rasterize_points.cu:265-330) -> radii[P] int32."""
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    radii = torch.zeros((P,), dtype=torch.int32, device=dev)
    radii_xy = torch.zeros((2 * P,), dtype=torch.int32, device=dev)
    if P != 0:
        m3, sc, rot = _f32(means3D, "means3D"), _f32(scales, "scales"), _f32(rotations, "rotations")
        vm, beams = _f32(viewmatrix, "viewmatrix"), _f32(beam_inclinations, "beam_inclinations")
        with torch.cuda.device(dev):
            rc = _lib.lidargs_surfel_visible_filter(
                None, None, None, None, None, None, C.c_int(P), C.c_int(0), C.c_int(int(image_width)), C.c_int(int(image_height)),
                _ptr(m3), _dev_or_none(sc), C.c_float(float(scale_modifier)), _dev_or_none(rot), None, _ptr(vm), None, _ptr(beams),
                C.c_int(int(bool(prefiltered))), C.c_int(int(lidar_far)), C.c_int(int(lidar_near)),
                _ptr(radii), _ptr(radii_xy), C.c_int(int(bool(debug))), _stream(dev))
        if rc < 0:
            _raise(rc, "rasterize_aussians_filter (surfel)")
    return radii


mark_visible = _base.mark_visible   # identical kernel in both variants (R2/cr/rasterizer_impl.cu:156-170)
