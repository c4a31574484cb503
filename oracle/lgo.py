"""ctypes/numpy front-end of the CPU oracle (oracle/lidargs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under lidar-gs_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblidargs_oracle.so")
_lib = None

_ARRAYS = {
    "depths": (0, np.float32), "means2D": (1, np.float32), "cov3D": (2, np.float32),
    "conic_opacity": (3, np.float32), "basis_u1": (4, np.float32), "basis_u2": (5, np.float32),
    "sphere": (6, np.float32), "tiles_touched": (7, np.uint32), "point_offsets": (8, np.uint32),
    "radii_xy": (9, np.int32), "keys": (10, np.uint64), "point_list": (11, np.uint32),
    "ranges": (12, np.uint32), "final_T": (13, np.float32), "n_contrib": (14, np.uint32),
}


def build(force=False):
    """Compile the oracle with gcc (seconds)."""
    srcs = [os.path.join(_HERE, f) for f in ("lidargs_oracle.c", "lidargs_surfel_oracle.c", "lgo_bench.c", "Makefile")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liblidargs_oracle.so"])
    return _SO


def _restypes(L):
    L.lgo_forward.restype = C.c_void_p
    L.lgo_forward_ex.restype = C.c_void_p
    L.lgo_backward_ex.restype = C.c_int
    L.sfo_forward.restype = C.c_void_p
    L.sfo_forward_tm.restype = C.c_void_p
    L.sfo_state_array.restype = C.c_void_p
    L.sfo_last_error.restype = C.c_char_p
    L.lgo_state_array.restype = C.c_void_p
    L.lgo_last_error.restype = C.c_char_p
    L.lgo_num_rendered.restype = C.c_int
    L.lgo_backward.restype = C.c_int
    L.lgo_bench_frames.restype = C.c_double
    L.sfo_num_rendered.restype = C.c_int
    return L


_lib_fma = None


class fma_build:
    """`with lgo.fma_build(): ...` -- inside, every oracle call runs the SAME restatement compiled with multiply-add contraction
    (-ffp-contract=fast -mfma: what nvcc's default -fmad=true does to the reference): a second conforming evaluation of the reference
    source, for the band of tests/util.py oracle_envelope.  Unavailable (host without FMA): the block runs on the plain build."""

    def __enter__(self):
        global _lib, _lib_fma
        lib()
        self.saved = _lib
        if _lib_fma is None:
            so = os.path.join(_HERE, "liblidargs_oracle_fma.so")
            try:
                subprocess.check_call(["make", "-s", "-C", _HERE, "liblidargs_oracle_fma.so"])
                _lib_fma = _restypes(C.CDLL(so))
            except Exception:
                _lib_fma = False
        if _lib_fma:
            _lib = _lib_fma
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.lgo_forward.restype = C.c_void_p
        _lib.lgo_forward_ex.restype = C.c_void_p
        _lib.lgo_backward_ex.restype = C.c_int
        _lib.sfo_forward.restype = C.c_void_p
        _lib.sfo_forward_tm.restype = C.c_void_p
        _lib.sfo_state_array.restype = C.c_void_p
        _lib.sfo_last_error.restype = C.c_char_p
        _lib.lgo_state_array.restype = C.c_void_p
        _lib.lgo_last_error.restype = C.c_char_p
        _lib.lgo_num_rendered.restype = C.c_int
        _lib.lgo_backward.restype = C.c_int
        _lib.lgo_bench_frames.restype = C.c_double
        _lib.sfo_num_rendered.restype = C.c_int
    return _lib


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.c_void_p)


class ForwardResult:
    """Outputs + the forward state the backward needs (freed on garbage collection)."""

    def __init__(self, handle, color, depth, occ, radii, inputs):
        self._h = handle
        self.color, self.depth, self.occ, self.radii = color, depth, occ, radii
        self.inputs = inputs
        self.num_rendered = lib().lgo_num_rendered(C.c_void_p(handle)) if handle else 0

    def array(self, name):
        which, dt = _ARRAYS[name]
        if not self._h:
            return np.zeros(0, dtype=dt)
        n = C.c_longlong(0)
        ptr = lib().lgo_state_array(C.c_void_p(self._h), which, C.byref(n))
        if n.value == 0:
            return np.zeros(0, dtype=dt)
        buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().lgo_free(C.c_void_p(self._h))
            self._h = None


def forward(means3D, colors, opacities, scales, rotations, viewmatrix, beams, W, H, bg=None,
            scale_modifier=1.0, cov3D_precomp=None, far=80, near=0, shell=None, T_in=None, t_only=False):
    """Restates CudaRasterizer::Rasterizer::forward (R3/cr/rasterizer_impl.cu:202-359).

    shell=(lo, hi), T_in, t_only: the multi-GPU range-shell extension (no reference counterpart);
    the result then carries .T_pass."""
    means3D = _f32(means3D); colors = _f32(colors); opacities = _f32(opacities)
    scales = _f32(scales); rotations = _f32(rotations); cov3D_precomp = _f32(cov3D_precomp)
    vm = _f32(viewmatrix).reshape(16); beams = _f32(beams)
    bg = _f32(np.zeros(2) if bg is None else bg)
    P = means3D.shape[0]
    color = np.zeros((2, H, W), np.float32); depth = np.zeros((1, H, W), np.float32)
    occ = np.zeros((1, H, W), np.float32); radii = np.zeros(P, np.int32)
    zero3 = np.zeros(16, np.float32)
    inputs = dict(means3D=means3D, colors=colors, opacities=opacities, scales=scales, rotations=rotations,
                  vm=vm, beams=beams, bg=bg, W=W, H=H, scale_modifier=scale_modifier,
                  cov3D_precomp=cov3D_precomp, far=far, near=near)
    if P == 0:      # the reference BINDING never enters the core for P == 0 (R3/rasterize_points.cu:87): all-zero outputs
        return ForwardResult(None, color, depth, occ, radii, inputs)
    if shell is None and T_in is None and not t_only:
        h = lib().lgo_forward(
            C.c_int(P), C.c_int(1), C.c_int(0), _p(bg), C.c_int(W), C.c_int(H),
            _p(means3D), None, _p(colors), _p(opacities), _p(scales), C.c_float(scale_modifier), _p(rotations),
            _p(cov3D_precomp), _p(vm), _p(zero3), _p(zero3), _p(beams), C.c_int(0), C.c_int(far), C.c_int(near),
            _p(color), _p(depth), _p(occ), _p(radii) if P else None)
        T_pass = None
    else:
        lo, hi = shell if shell is not None else (-np.inf, np.inf)
        T_in = None if T_in is None else _f32(T_in).reshape(-1)
        T_pass = np.zeros(H * W, np.float32)
        h = lib().lgo_forward_ex(
            C.c_int(P), C.c_int(1), C.c_int(0), _p(bg), C.c_int(W), C.c_int(H),
            _p(means3D), None, _p(colors), _p(opacities), _p(scales), C.c_float(scale_modifier), _p(rotations),
            _p(cov3D_precomp), _p(vm), _p(zero3), _p(zero3), _p(beams), C.c_int(0), C.c_int(far), C.c_int(near),
            C.c_float(lo), C.c_float(hi), _p(T_in), C.c_int(int(t_only)), _p(T_pass),
            _p(color), _p(depth), _p(occ), _p(radii) if P else None)
    if not h:
        raise RuntimeError(lib().lgo_last_error().decode())
    res = ForwardResult(h, color, depth, occ, radii, inputs)
    res.T_pass = T_pass
    return res


def render_shell(fwd, T_in=None, t_only=False, bg=None):
    """Phase 2 of the two-phase shell render on an already-binned oracle state."""
    i = fwd.inputs
    H, W = i["H"], i["W"]
    T_in = None if T_in is None else _f32(T_in).reshape(-1)
    bg = None if bg is None else _f32(bg)
    T_pass = np.zeros(H * W, np.float32)
    lib().lgo_render_ex(C.c_void_p(fwd._h), _p(i["colors"]), _p(bg), _p(i["beams"]), _p(T_in), C.c_int(int(t_only)),
                        _p(fwd.color), _p(fwd.depth), _p(fwd.occ), _p(T_pass))
    fwd.T_pass = T_pass
    return fwd


def backward(fwd, dL_dcolor, dL_ddepth, dL_docc, behind=None, T_final_global=None, bg=None):
    """Restates CudaRasterizer::Rasterizer::backward (R3/cr/rasterizer_impl.cu:431-549).

    Returns a dict with the 8 tensors the reference binding returns
    (R3/rasterize_points.cu:218) plus the 5 internal per-Gaussian accumulators."""
    i = fwd.inputs
    P, W, H = i["means3D"].shape[0], i["W"], i["H"]
    dL_dcolor = _f32(dL_dcolor).reshape(2, H, W); dL_ddepth = _f32(dL_ddepth).reshape(H, W)
    dL_docc = _f32(dL_docc).reshape(H, W)
    z = lambda *s: np.zeros(s, np.float32)
    g = dict(dL_dmeans2D=z(P, 4), dL_dconic=z(P, 4), dL_dopacity=z(P, 1), dL_dcolors=z(P, 2), dL_ddepths=z(P, 1),
             dL_dmeans3D=z(P, 3), dL_dsphere=z(P, 3), dL_dbasis_u1=z(P, 3), dL_dbasis_u2=z(P, 3),
             dL_dcov3D=z(P, 6), dL_dsh=z(P, 0, 3), dL_dscales=z(P, 3), dL_drotations=z(P, 4))
    zero3 = np.zeros(16, np.float32)
    if P == 0:      # R3/rasterize_points.cu:177
        return g
    behind = None if behind is None else _f32(behind).reshape(-1)
    T_final_global = None if T_final_global is None else _f32(T_final_global).reshape(-1)
    rc = lib().lgo_backward_ex(
        C.c_void_p(fwd._h), C.c_int(P), C.c_int(1), C.c_int(0), C.c_int(fwd.num_rendered), _p(i["bg"] if bg is None else _f32(bg)),
        C.c_int(W), C.c_int(H), _p(i["means3D"]), None, _p(i["colors"]), _p(i["scales"]),
        C.c_float(i["scale_modifier"]), _p(i["rotations"]), _p(i["cov3D_precomp"]), _p(i["vm"]), _p(zero3), _p(zero3),
        _p(i["beams"]), C.c_float(1.0), C.c_float(1.0), _p(fwd.radii),
        _p(dL_dcolor), _p(dL_ddepth), _p(dL_docc),
        _p(g["dL_dmeans2D"]), _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_ddepths"]),
        _p(g["dL_dmeans3D"]), _p(g["dL_dsphere"]), _p(g["dL_dbasis_u1"]), _p(g["dL_dbasis_u2"]),
        _p(g["dL_dcov3D"]), None, _p(g["dL_dscales"]), _p(g["dL_drotations"]), _p(behind), _p(T_final_global))
    if rc != 0:
        raise RuntimeError(lib().lgo_last_error().decode())
    return g


def visible_filter(means3D, scales, rotations, viewmatrix, beams, W, H, scale_modifier=1.0,
                   cov3D_precomp=None, far=80, near=0):
    """Restates Rasterizer::visible_filter (R3/cr/rasterizer_impl.cu:362-426) -> radii."""
    means3D = _f32(means3D); scales = _f32(scales); rotations = _f32(rotations); cov3D_precomp = _f32(cov3D_precomp)
    vm = _f32(viewmatrix).reshape(16); beams = _f32(beams)
    P = means3D.shape[0]
    radii = np.zeros(P, np.int32)
    zero3 = np.zeros(16, np.float32)
    lib().lgo_visible_filter(C.c_int(P), C.c_int(0), C.c_int(W), C.c_int(H), _p(means3D), _p(scales),
                             C.c_float(scale_modifier), _p(rotations), _p(cov3D_precomp), _p(vm), _p(zero3), _p(zero3),
                             _p(beams), C.c_float(1.0), C.c_float(1.0), C.c_int(0), C.c_int(far), C.c_int(near),
                             _p(radii))
    return radii


def rects(p_cr, r_xy, tiles_x, tiles_y, surfel=False):
    """getRect_lidar of either variant on caller-supplied (p.x, p.y), (rx, ry): int32 [n, 4] = (xmin, ymin, xmax, ymax)."""
    p = np.ascontiguousarray(p_cr, dtype=np.float32).reshape(-1, 2)
    r = np.ascontiguousarray(r_xy, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((p.shape[0], 4), np.int32)
    fn = lib().sfo_rects if surfel else lib().lgo_rects
    fn(C.c_int(p.shape[0]), p.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), C.c_int(tiles_x), C.c_int(tiles_y),
       out.ctypes.data_as(C.c_void_p))
    return out


def pixel_dirs(W, H, beams):
    """[H,W,3] unit ray of every pixel as the blend kernels evaluate it (R3/cr/forward.cu:589-591)."""
    beams = _f32(beams)
    out = np.zeros((H, W, 3), np.float32)
    tmp = np.zeros(3, np.float32)
    L = lib()
    for y in range(H):
        for x in range(W):
            L.lgo_pixel_dir(C.c_int(x), C.c_int(y), C.c_int(W), C.c_int(H), _p(beams), _p(tmp))
            out[y, x] = tmp
    return out


def mark_visible(means3D, viewmatrix):
    """Restates Rasterizer::markVisible (R3/cr/rasterizer_impl.cu:142-154)."""
    means3D = _f32(means3D); vm = _f32(viewmatrix).reshape(16)
    P = means3D.shape[0]
    present = np.zeros(P, np.uint8)
    lib().lgo_mark_visible(C.c_int(P), _p(means3D), _p(vm), None, _p(present))
    return present.astype(bool)


def bench_frames(scene, W, H, grads, threads, frames_each, fwd_only=False, surfel=False, far=80, near=0):
    """Wall seconds for threads x frames_each frames rendered by `threads` POSIX threads (oracle/lgo_bench.c): the all-core
    CPU baseline of bench.py.  grads = (colour, depth, occ) or, surfel, (colour, others)."""
    a = {k: _f32(scene[k]) for k in ("bg", "means3D", "colors", "opacities", "scales", "rotations", "beams")}
    vm = _f32(scene["viewmatrix"]).reshape(16)
    g = [_f32(x).reshape(-1) for x in grads] + [None]
    s = lib().lgo_bench_frames(C.c_int(int(surfel)), C.c_int(threads), C.c_int(frames_each), C.c_int(int(fwd_only)),
                               C.c_int(a["means3D"].shape[0]), C.c_int(W), C.c_int(H), C.c_int(far), C.c_int(near), _p(a["bg"]), _p(a["means3D"]),
                               _p(a["colors"]), _p(a["opacities"]), _p(a["scales"]), _p(a["rotations"]), _p(vm), _p(a["beams"]),
                               _p(g[0]), _p(g[1]), _p(g[2]) if g[2] is not None else None)
    if s < 0:
        raise RuntimeError("lgo_bench_frames: a frame failed: " + lib().lgo_last_error().decode())
    return float(s)
