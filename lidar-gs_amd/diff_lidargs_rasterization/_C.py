"""`diff_lidargs_rasterization._C` -- the native module of the drop-in package.

The reference's `_C` is a pybind11/torch extension (R3/ext.cpp:15-21) wrapping
R3/rasterize_points.cu.  This one binds the same four functions, with the same names,
positional signatures, return tuples and error behaviour, to the C ABI declared in
include/lidargs_rasterizer.h (liblidargs_hip.so, hand-written HIP for gfx950) through ctypes.
torch is used for exactly what R3/rasterize_points.cu uses it for: allocating the outputs, the
three resizable scratch byte tensors and the zero-filled gradient tensors, and handing over raw
device pointers plus the current HIP stream.

There is NO CPU path and no fallback: tensors must live on a HIP device ("cuda" in PyTorch-ROCm)
and the shared library must have been built (python lidar-gs_amd/build_hip.py, or
__graft_entry__.build()); otherwise importing or calling this module fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO_PATH = os.path.join(_HERE, "liblidargs_hip.so")
NUM_CHANNELS = 2  # R3/cr/config.h:15

if not os.path.exists(_SO_PATH):
    raise ImportError(
        f"diff_lidargs_rasterization: native library {_SO_PATH} is missing. Build it with "
        "`python lidar-gs_amd/build_hip.py` (hipcc --offload-arch=gfx950). There is no CPU fallback.")

_lib = C.CDLL(_SO_PATH)
_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_lib.lidargs_last_error.restype = C.c_char_p
for _name in ("lidargs_forward", "lidargs_forward_enqueue", "lidargs_backward", "lidargs_visible_filter", "lidargs_mark_visible",
              "lidargs_forward_shell", "lidargs_render_shell", "lidargs_backward_shell", "lidargs_abi_version",
              "lidargs_profile_read", "lidargs_profile_summary", "lidargs_last_counters"):
    getattr(_lib, _name).restype = C.c_int
_lib.lidargs_profile_stage_name.restype = C.c_char_p
_lib.lidargs_profile_enable.restype = None
_lib.lidargs_counters_enable.restype = None
if _lib.lidargs_abi_version() != 2:
    raise ImportError("diff_lidargs_rasterization: liblidargs_hip.so ABI version mismatch; rebuild it")


def _err():
    return _lib.lidargs_last_error().decode(errors="replace")


def _raise(code, what):
    msg = _err()
    if code == -2:  # LIDARGS_ERR_NO_COLORS: the reference throws std::runtime_error here (R3/cr/rasterizer_impl.cu:249-252)
        raise RuntimeError(msg)
    raise RuntimeError(f"{what} failed with code {code}: {msg}")


def _require_device(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"diff_lidargs_rasterization: `{name}` must be a tensor on a HIP device (device='cuda'); "
            "this build has no CPU rasterizer and does not fall back to one")


def _f32(t, name):
    """`.contiguous().data<float>()` of the reference binding: float32 required, made contiguous."""
    if t.numel() and t.dtype != torch.float32:
        raise RuntimeError(f"expected scalar type Float but found {t.dtype} for `{name}`")
    return t.contiguous()


def _ptr(t):
    """Device pointer, NULL for empty tensors (torch gives data_ptr()==0 for them, which is what the
    reference relies on for `cov3D_precomp != nullptr`, R3/cr/forward.cu:307)."""
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


_scratch_registry = {}
# LIDARGS_POISON_SCRATCH=1 (tests): every scratch buffer handed to the library is filled with 0xFF bytes (NaN floats, huge
# integers) first, so that a kernel reading scratch memory nobody wrote shows up deterministically instead of depending on what
# the caching allocator happens to recycle.
_POISON = os.environ.get("LIDARGS_POISON_SCRATCH", "0") == "1"


@_ALLOC_FN
def _alloc_cb(user, nbytes):
    """The one allocator callback handed to the C ABI; `user` is the key of a live _Scratch."""
    sc = _scratch_registry.get(user)
    if sc is None:
        return 0
    try:
        sc.tensor = torch.empty(int(nbytes), dtype=torch.uint8, device=sc.device)
        if _POISON:
            sc.tensor.fill_(255)
        return sc.tensor.data_ptr()
    except Exception:  # surface as LIDARGS_ERR_ALLOC instead of unwinding through C
        return 0


class _Scratch:
    """One resizable byte tensor = resizeFunctional(t) of R3/rasterize_points.cu:27-33.

    No per-call ctypes closure (a closure over `self` would form a reference cycle and keep the
    hundreds-of-MB scratch tensor alive until Python's cyclic GC runs -> one hipMalloc per frame)."""
    __slots__ = ("device", "tensor", "key")
    cb = _alloc_cb

    def __init__(self, device):
        self.device = device
        self.tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.key = id(self)
        _scratch_registry[self.key] = self

    @property
    def user(self):
        return C.c_void_p(self.key)

    def take(self):
        """Detach from the registry and hand the tensor over."""
        _scratch_registry.pop(self.key, None)
        return self.tensor

    def __del__(self):
        _scratch_registry.pop(getattr(self, "key", None), None)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, image_height, image_width, beam_inclinations, sh, degree, campos,
                        prefiltered, far, near, debug, enqueue=None):
    """RasterizeGaussiansCUDA (R3/rasterize_points.cu:35-124).

    Returns (num_rendered, out_color[2,H,W], out_depth[1,H,W], out_occ[1,H,W], radii[P] int32,
    geomBuffer, binningBuffer, imgBuffer) -- the last three opaque uint8 tensors.

    enqueue = (instance_capacity, tile_rows, status) switches to lidargs_forward_enqueue (no host wait, graph-capturable;
    include/lidargs_rasterizer.h): `status` is None or a pinned int32[16] host tensor the stream fills behind the launches."""
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_device(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    # the reference fills these with zeros (R3/rasterize_points.cu:63-68) and its kernels overwrite them; the native forward
    # writes every pixel and every radii row, so for P > 0 five fill launches are saved by not pre-zeroing
    mk = torch.empty if P != 0 else torch.zeros
    out_color = mk((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_depth = mk((1, H, W), dtype=torch.float32, device=dev)
    out_occ = mk((1, H, W), dtype=torch.float32, device=dev)
    radii = mk((P,), dtype=torch.int32, device=dev)
    radii_xy = None           # the reference's binding makes this [2P] tensor and never returns it (R3/rasterize_points.cu:75): NULL = not written
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    rendered = 0
    if P != 0:
        M = int(sh.size(1)) if sh.numel() != 0 and sh.ndim > 1 else 0
        bg, m3, col, opa = _f32(background, "background"), _f32(means3D, "means3D"), _f32(colors, "colors"), _f32(opacity, "opacity")
        sc, rot, cov = _f32(scales, "scales"), _f32(rotations, "rotations"), _f32(cov3D_precomp, "cov3D_precomp")
        vm, pm, cp = _f32(viewmatrix, "viewmatrix"), _f32(projmatrix, "projmatrix"), _f32(campos, "campos")
        beams, shc = _f32(beam_inclinations, "beam_inclinations"), _f32(sh, "sh")
        for t, n in ((bg, "bg"), (col, "colors_precomp"), (opa, "opacities"), (vm, "viewmatrix"), (beams, "beam_inclinations")):
            if t.numel():
                _require_device(t, n)
        common = (_alloc_cb, geom.user, _alloc_cb, binning.user, _alloc_cb, img.user,
                  C.c_int(P), C.c_int(int(degree)), C.c_int(M), _ptr(bg), C.c_int(W), C.c_int(H),
                  _ptr(m3), _ptr(shc if shc.is_cuda else None), _ptr(col), _ptr(opa), _ptr(sc if sc.is_cuda else None),
                  C.c_float(float(scale_modifier)), _ptr(rot if rot.is_cuda else None), _ptr(cov if cov.is_cuda else None),
                  _ptr(vm), _ptr(pm if pm.is_cuda else None), _ptr(cp if cp.is_cuda else None), _ptr(beams),
                  C.c_int(int(bool(prefiltered))), C.c_int(int(far)), C.c_int(int(near)),
                  _ptr(out_color), _ptr(out_depth), _ptr(out_occ), _ptr(radii), _ptr(radii_xy), C.c_int(int(bool(debug))))
        with torch.cuda.device(dev):
            if enqueue is None:
                rendered = _lib.lidargs_forward(*common, _stream(dev))
            else:
                cap, tile_rows, status = enqueue
                if status is not None and not (status.dtype == torch.int32 and status.numel() >= 16 and status.is_pinned()):
                    raise RuntimeError("enqueue status must be a pinned int32 host tensor of at least 16 elements")
                rendered = _lib.lidargs_forward_enqueue(*common, C.c_int(int(cap)), C.c_int(int(tile_rows)),
                                                        C.c_void_p(status.data_ptr()) if status is not None else None, _stream(dev))
        if rendered < 0:
            geom.take(); binning.take(); img.take()
            _raise(rendered, "rasterize_gaussians")
    return rendered, out_color, out_depth, out_occ, radii, geom.take(), binning.take(), img.take()


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, beam_inclinations, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_depth, dL_dout_occ, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                 debug, want_cov3D_grad=True):
    """RasterizeGaussiansBackwardCUDA (R3/rasterize_points.cu:126-219).

    Returns (dL_dmeans2D[P,4], dL_dcolors[P,2], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6],
    dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4]).  want_cov3D_grad=False (only with scales + rotations, where the
    reference's dL_dcov3D is an intermediate nobody receives): dL_dcov3D is not materialised and None is returned for it."""
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 and sh.ndim > 1 else 0
    # the reference makes 13 torch::zeros tensors (:163-175), five of which are scratch it never returns.  Here: one slab for
    # the returned ones, NOT pre-zeroed (the native backward writes every row of every output, zeros for culled
    # Gaussians), and NULL for the scratch ones, which the library then does not materialise.
    skip_cov = (not want_cov3D_grad) and cov3D_precomp.numel() == 0
    widths = (3, 4, NUM_CHANNELS, 1, 0 if skip_cov else 6, 3, 4)
    slab = torch.empty(P * sum(widths), dtype=torch.float32, device=dev)
    parts, o = [], 0
    for w in widths:
        parts.append(slab[o:o + P * w].view(P, w)); o += P * w
    dL_dmeans3D, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dcov3D, dL_dscales, dL_drotations = parts
    if skip_cov:
        dL_dcov3D = None
    dL_ddepths = dL_dconic = dL_dsphere = dL_du1 = dL_du2 = None
    dL_dsh = torch.zeros((P, M, 3), dtype=torch.float32, device=dev)
    if P != 0:
        bg, m3, col = _f32(background, "background"), _f32(means3D, "means3D"), _f32(colors, "colors")
        cov, vm, beams = _f32(cov3D_precomp, "cov3D_precomp"), _f32(viewmatrix, "viewmatrix"), _f32(beam_inclinations, "beam_inclinations")
        g0, g1, g2 = _f32(dL_dout_color, "dL_dout_color"), _f32(dL_dout_depth, "dL_dout_depth"), _f32(dL_dout_occ, "dL_dout_occ")
        rad = radii.contiguous()
        # the reference passes scales/rotations by raw data_ptr here (:185,:187); contiguity is the caller's business there,
        # we make it explicit instead of silently reading a strided tensor wrongly
        sc, rot = _f32(scales, "scales"), _f32(rotations, "rotations")
        with torch.cuda.device(dev):
            rc = _lib.lidargs_backward(
                C.c_int(P), C.c_int(int(degree)), C.c_int(M), C.c_int(int(R)), _ptr(bg), C.c_int(W), C.c_int(H),
                _ptr(m3), None, _ptr(col), _ptr(sc if sc.is_cuda else None), C.c_float(float(scale_modifier)),
                _ptr(rot if rot.is_cuda else None), _ptr(cov if cov.is_cuda else None), _ptr(vm), None, None, _ptr(beams),
                C.c_float(float(tan_fovx)), C.c_float(float(tan_fovy)), _ptr(rad),
                _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(g0), _ptr(g1), _ptr(g2),
                _ptr(dL_dmeans2D), _ptr(dL_dconic), _ptr(dL_dopacity), _ptr(dL_dcolors), _ptr(dL_ddepths), _ptr(dL_dmeans3D),
                _ptr(dL_dsphere), _ptr(dL_du1), _ptr(dL_du2), _ptr(dL_dcov3D), None, _ptr(dL_dscales), _ptr(dL_drotations),
                C.c_int(int(bool(debug))), _stream(dev))
        if rc < 0:
            _raise(rc, "rasterize_gaussians_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def rasterize_aussians_filter(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, campos,
                              tan_fovx, tan_fovy, image_height, image_width, beam_inclinations, prefiltered, far, near, debug):
    """RasterizeGaussiansfilterCUDA (R3/rasterize_points.cu:243-318); the misspelt name is the reference's
    (R3/ext.cpp:18).  Returns radii[P] int32."""
    if means3D.ndim != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    radii = (torch.empty if P != 0 else torch.zeros)((P,), dtype=torch.int32, device=dev)   # every row is written by the library
    radii_xy = None           # (as in the forward: the reference never returns it, R3/rasterize_points.cu:278)
    if P != 0:
        m3, sc, rot = _f32(means3D, "means3D"), _f32(scales, "scales"), _f32(rotations, "rotations")
        cov, vm, beams = _f32(cov3D_precomp, "cov3D_precomp"), _f32(viewmatrix, "viewmatrix"), _f32(beam_inclinations, "beam_inclinations")
        with torch.cuda.device(dev):
            rc = _lib.lidargs_visible_filter(
                None, None, None, None, None, None, C.c_int(P), C.c_int(0), C.c_int(int(image_width)), C.c_int(int(image_height)),
                _ptr(m3), _ptr(sc if sc.is_cuda else None), C.c_float(float(scale_modifier)), _ptr(rot if rot.is_cuda else None),
                _ptr(cov if cov.is_cuda else None), _ptr(vm), None, None, _ptr(beams), C.c_float(float(tan_fovx)),
                C.c_float(float(tan_fovy)), C.c_int(int(bool(prefiltered))), C.c_int(int(far)), C.c_int(int(near)),
                _ptr(radii), _ptr(radii_xy), C.c_int(int(bool(debug))), _stream(dev))
        if rc < 0:
            _raise(rc, "rasterize_aussians_filter")
    return radii


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible (R3/rasterize_points.cu:221-240) -> bool[P]."""
    _require_device(means3D, "means3D")
    dev = means3D.device
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        m3, vm = _f32(means3D, "means3D"), _f32(viewmatrix, "viewmatrix")
        with torch.cuda.device(dev):
            rc = _lib.lidargs_mark_visible(C.c_int(P), _ptr(m3), _ptr(vm), None, _ptr(present), _stream(dev))
        if rc < 0:
            _raise(rc, "mark_visible")
    return present


# ---- introspection (bench.py / tests) ----------------------------------------------------------------------------
def profile_enable(on=True):
    """False/0: off; True/1: stage events on every call; N > 1: on every N-th forward and every N-th backward (an event costs
    ~4.5 us of device time, a dozen per call 6 % of a 1 ms frame)."""
    _lib.lidargs_profile_enable(C.c_int(int(on)))


def profile_read():
    """[(stage name, milliseconds)] of the last forward or backward call on this thread."""
    buf = (C.c_float * 24)()
    n = _lib.lidargs_profile_read(buf, C.c_int(24))
    return [(_lib.lidargs_profile_stage_name(C.c_int(i)).decode(), float(buf[i])) for i in range(n)]


def profile_summary():
    """{stage name: (mean milliseconds, samples)} over every call recorded since profile_enable(True)."""
    names = (C.c_char_p * 48)()
    tot = (C.c_float * 48)()
    cnt = (C.c_int * 48)()
    n = _lib.lidargs_profile_summary(names, tot, cnt, C.c_int(48))
    return {names[i].decode(): (float(tot[i]) / max(1, cnt[i]), int(cnt[i])) for i in range(n)}


def counters_enable(on):
    """While on, every forward of this thread ends with the small counting launches behind V, R_ref, taken_instances, touched and
    backward_entries of last_counters() (diagnostics: keep it off inside timed regions)."""
    _lib.lidargs_counters_enable(C.c_int(1 if on else 0))


def last_counters():
    """dict(P, V, instances, R_ref, tile_rows, tiles, ..., touched) of the last forward on this thread; the device-side ones (V, R_ref,
    taken_instances, touched, backward_entries) are -1 unless counters_enable(True) was in force during that forward."""
    buf = (C.c_longlong * 10)()
    _lib.lidargs_last_counters(buf, C.c_int(10))
    return dict(P=buf[0], V=buf[1], instances=buf[2], R_ref=buf[3], tile_rows=buf[4], tiles=buf[5], taken_instances=buf[6],
                segments=buf[7], touched=buf[8], backward_entries=buf[9])
