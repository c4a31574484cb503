"""ctypes/numpy front-end of the surfel oracle (oracle/lidargs_surfel_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from .lgo import _f32, _p, lib

_ARRAYS = {"depths": (0, np.float32), "means2D": (1, np.float32), "transMat": (2, np.float32), "normal_opacity": (3, np.float32),
           "tiles_touched": (4, np.uint32), "radii_xy": (5, np.int32), "point_list": (6, np.uint32), "ranges": (7, np.uint32),
           "accum": (8, np.float32), "n_contrib": (9, np.uint32)}


class SurfelForward:
    def __init__(self, h, color, others, radii, inputs):
        self._h, self.color, self.others, self.radii, self.inputs = h, color, others, radii, inputs
        self.num_rendered = lib().sfo_num_rendered(C.c_void_p(h)) if h else 0

    def array(self, name):
        which, dt = _ARRAYS[name]
        if not self._h:
            return np.zeros(0, dt)
        n = C.c_longlong(0)
        ptr = lib().sfo_state_array(C.c_void_p(self._h), which, C.byref(n))
        if n.value == 0:
            return np.zeros(0, dt)
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sfo_free(C.c_void_p(self._h)); self._h = None


def forward(means3D, colors, opacities, scales2, rotations, viewmatrix, beams, W, H, bg=None, scale_modifier=1.0, far=80, near=0,
            transMat_precomp=None):
    """Restates R2 Rasterizer::forward (R2/cr/rasterizer_impl.cu:200-360): color[2,H,W], others[7,H,W], radii[P].
    transMat_precomp [P,9] (optional): the rows the blends use instead of the ones built from scales / rotations (:332, :408)."""
    means3D = _f32(means3D); colors = _f32(colors); opacities = _f32(opacities); scales2 = _f32(scales2); rotations = _f32(rotations)
    vm = _f32(viewmatrix).reshape(16); beams = _f32(beams); bg = _f32(np.zeros(2) if bg is None else bg)
    P = means3D.shape[0]
    color = np.zeros((2, H, W), np.float32); others = np.zeros((7, H, W), np.float32); radii = np.zeros(P, np.int32)
    inputs = dict(means3D=means3D, colors=colors, opacities=opacities, scales=scales2, rotations=rotations, vm=vm, beams=beams, bg=bg,
                  W=W, H=H, scale_modifier=scale_modifier)
    if P == 0:
        return SurfelForward(None, color, others, radii, inputs)
    tm = None if transMat_precomp is None else _f32(transMat_precomp).reshape(P, 9)
    h = lib().sfo_forward_tm(C.c_int(P), _p(bg), C.c_int(W), C.c_int(H), _p(means3D), _p(colors), _p(opacities), _p(scales2),
                             C.c_float(scale_modifier), _p(rotations), _p(tm), _p(vm), _p(beams), C.c_int(far),
                             C.c_int(near), _p(color), _p(others), _p(radii))
    if not h:
        raise RuntimeError(lib().sfo_last_error().decode())
    return SurfelForward(h, color, others, radii, inputs)


def backward(fwd, dL_dcolor, dL_dothers):
    """Restates R2 Rasterizer::backward; returns the 9 tensors of R2/rasterize_points.cu:241 plus intermediates."""
    i = fwd.inputs
    P, W, H = i["means3D"].shape[0], i["W"], i["H"]
    dL_dcolor = _f32(dL_dcolor).reshape(2, H, W); dL_dothers = _f32(dL_dothers).reshape(7, H, W)
    z = lambda *s: np.zeros(s, np.float32)
    g = dict(dL_dmeans2D=z(P, 4), dL_dnormal=z(P, 3), dL_dopacity=z(P, 1), dL_dcolors=z(P, 2), dL_dmeans3D=z(P, 3), dL_dtransMat=z(P, 9),
             dL_dtransMat_2dtemp=z(P, 3), dL_dscales=z(P, 2), dL_drotations=z(P, 4), depth=z(P, 1))
    if P == 0:
        return g
    rc = lib().sfo_backward(C.c_void_p(fwd._h), C.c_int(P), C.c_int(fwd.num_rendered), _p(i["bg"]), C.c_int(W), C.c_int(H), _p(i["means3D"]),
                            _p(i["colors"]), _p(i["scales"]), C.c_float(i["scale_modifier"]), _p(i["rotations"]), _p(i["vm"]), _p(i["beams"]),
                            _p(fwd.radii), _p(dL_dcolor), _p(dL_dothers), _p(g["dL_dmeans2D"]), _p(g["dL_dnormal"]), _p(g["dL_dopacity"]),
                            _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]), _p(g["dL_dtransMat"]), _p(g["dL_dtransMat_2dtemp"]), _p(g["dL_dscales"]),
                            _p(g["dL_drotations"]), _p(g["depth"]))
    if rc != 0:
        raise RuntimeError(lib().sfo_last_error().decode())
    return g


def visible_filter(means3D, scales2, rotations, viewmatrix, beams, W, H, scale_modifier=1.0, far=80, near=0):
    means3D = _f32(means3D); scales2 = _f32(scales2); rotations = _f32(rotations); vm = _f32(viewmatrix).reshape(16); beams = _f32(beams)
    P = means3D.shape[0]
    radii = np.zeros(P, np.int32)
    lib().sfo_visible_filter(C.c_int(P), C.c_int(W), C.c_int(H), _p(means3D), _p(scales2), C.c_float(scale_modifier), _p(rotations), _p(vm),
                             _p(beams), C.c_int(far), C.c_int(near), _p(radii))
    return radii
