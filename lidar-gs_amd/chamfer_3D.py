"""`chamfer_3D` -- drop-in for the reference's compiled chamfer extension (extern/chamfer3D/chamfer_cuda.cpp + chamfer3D.cu).

extern/chamfer3D/dist_chamfer_3D.py looks for a module of this name first (`importlib.find_loader("chamfer_3D")`, :14) and only
JIT-compiles its CUDA sources when it is missing; with lidar-gs_amd/ on PYTHONPATH the reference's own `chamfer_3DDist` wrapper
therefore runs on the HIP kernels of include/lidargs_chamfer.h unchanged.  Same two functions, same in-place contract."""
import ctypes as C

import torch

from diff_lidargs_rasterization import _C as _base

_lib = _base._lib
_lib.lidargs_chamfer_forward.restype = C.c_int
_lib.lidargs_chamfer_backward.restype = C.c_int
_lib.lidargs_chamfer_scratch_bytes.restype = C.c_size_t


def _chk(t, name, dtype):
    _base._require_device(t, name)
    if t.dtype != dtype or not t.is_contiguous():
        raise RuntimeError(f"chamfer_3D: `{name}` must be a contiguous {dtype} tensor")


def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
    """chamfer_cuda_forward (chamfer3D.cu:141-166): fills dist1 [B,n], dist2 [B,m] (squared distances) and idx1, idx2 (int32) in place."""
    for t, n in ((xyz1, "xyz1"), (xyz2, "xyz2"), (dist1, "dist1"), (dist2, "dist2")):
        _chk(t, n, torch.float32)
    _chk(idx1, "idx1", torch.int32); _chk(idx2, "idx2", torch.int32)
    B, n, m = int(xyz1.shape[0]), int(xyz1.shape[1]), int(xyz2.shape[1])
    p = _base._ptr
    nb = int(_lib.lidargs_chamfer_scratch_bytes(C.c_int(B), C.c_int(n), C.c_int(m)))
    scratch = torch.empty(nb, dtype=torch.uint8, device=xyz1.device)
    with torch.cuda.device(xyz1.device):
        rc = _lib.lidargs_chamfer_forward(C.c_int(B), C.c_int(n), C.c_int(m), p(xyz1), p(xyz2), p(dist1), p(dist2), p(idx1), p(idx2),
                                          p(scratch), C.c_size_t(nb), _base._stream(xyz1.device))
    if rc < 0:
        _base._raise(rc, "chamfer_3D.forward")
    return 1


def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
    """chamfer_cuda_backward (chamfer3D.cu:196-226): accumulates into gradxyz1 / gradxyz2 (zero-initialised by the caller)."""
    for t, n in ((xyz1, "xyz1"), (xyz2, "xyz2"), (gradxyz1, "gradxyz1"), (gradxyz2, "gradxyz2"), (graddist1, "graddist1"), (graddist2, "graddist2")):
        _chk(t, n, torch.float32)
    _chk(idx1, "idx1", torch.int32); _chk(idx2, "idx2", torch.int32)
    B, n, m = int(xyz1.shape[0]), int(xyz1.shape[1]), int(xyz2.shape[1])
    p = _base._ptr
    with torch.cuda.device(xyz1.device):
        rc = _lib.lidargs_chamfer_backward(C.c_int(B), C.c_int(n), C.c_int(m), p(xyz1), p(xyz2), p(graddist1), p(graddist2), p(idx1), p(idx2),
                                           p(gradxyz1), p(gradxyz2), _base._stream(xyz1.device))
    if rc < 0:
        _base._raise(rc, "chamfer_3D.backward")
    return 1
