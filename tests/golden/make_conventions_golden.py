"""Generate tests/golden/conventions_golden.npz by EXECUTING the reference's own Python for the two conventions every
rasterizer input rests on.  Run in the authoring container only (needs /root/reference):

    python tests/golden/make_conventions_golden.py

1. The view matrix.  `raster_settings.viewmatrix` is `viewpoint_camera.world_view_transform`
   (gaussian_renderer/__init__.py:157), built in scene/cameras.py:56 as
   `torch.tensor(getWorld2View2(R, T, trans, scale)).transpose(0, 1)` (+ `.cuda()`), with getWorld2View2 of
   utils/graphics_utils.py:38-49, and `camera_center = world_view_transform.inverse()[3, :3]` (scene/cameras.py:59).
   Executed here: the function (pulled out of the module's AST -- the module imports nothing exotic, but this keeps the
   recipe uniform), and the two right-hand sides of scene/cameras.py AS PARSED: the sub-expression under the trailing
   `.cuda()` call is compiled from the reference's own AST node (scene/cameras.py cannot be imported: it pulls in the
   dataset/CUDA stack).
2. The covariance the `cov3D_precomp` caller passes (gaussian_renderer/__init__.py:246-247 -> scene/gaussian_model.py:269-270
   -> :33-37 build_covariance_from_scaling_rotation -> utils/general_utils.py:65-113 strip_lowerdiag / strip_symmetric /
   build_rotation / build_scaling_rotation).  Those functions create their tensors with device="cuda"; their text is executed
   UNCHANGED under a TorchFunctionMode that re-targets factory calls from "cuda" to "cpu" (device placement only; the
   arithmetic is the reference's, in fp32 on CPU torch).

Only inputs and outputs are stored (float arrays); no reference source text enters the repository.
"""
import ast
import os

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"


class CudaToCpu(TorchFunctionMode):
    """device='cuda' -> 'cpu' for the factory calls of the executed reference functions (no GPU in this container)."""

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if str(kwargs.get("device", "")).startswith("cuda"):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def functions_of(path, names, ns):
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), [n.name for n in body]
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def camera_expressions():
    """The value expressions of `self.world_view_transform = ...` (without its trailing .cuda()) and `self.camera_center = ...`
    in Camera.__init__ of scene/cameras.py, as code objects compiled from the reference's AST."""
    path = os.path.join(REF, "scene", "cameras.py")
    tree = ast.parse(open(path).read(), path)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Attribute):
            name = node.targets[0].attr
            if name == "world_view_transform" and "world_view_transform" not in found:
                v = node.value
                assert isinstance(v, ast.Call) and isinstance(v.func, ast.Attribute) and v.func.attr == "cuda", ast.dump(v)[:200]
                found[name] = compile(ast.fix_missing_locations(ast.Expression(v.func.value)), path, "eval")
            elif name == "camera_center" and "camera_center" not in found:
                found[name] = compile(ast.fix_missing_locations(ast.Expression(node.value)), path, "eval")
    assert set(found) == {"world_view_transform", "camera_center"}, found.keys()
    return found


def gaussian_model_cov_function(ns):
    """build_covariance_from_scaling_rotation: the nested def of GaussianModel.setup_functions (scene/gaussian_model.py:33-37)."""
    path = os.path.join(REF, "scene", "gaussian_model.py")
    tree = ast.parse(open(path).read(), path)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "build_covariance_from_scaling_rotation":
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            return ns["build_covariance_from_scaling_rotation"]
    raise AssertionError("build_covariance_from_scaling_rotation not found")


def main():
    rng = np.random.default_rng(20260928)
    out = {}
    # ---- 1. view matrix ------------------------------------------------------------------------------------------------
    ns = functions_of(os.path.join(REF, "utils", "graphics_utils.py"), ["getWorld2View2"], {"np": np})
    exprs = camera_expressions()
    K = 6
    Rs, Ts, Vs, Cs, W2Vs = [], [], [], [], []
    for k in range(K):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        R = Q if k else np.eye(3)
        T = rng.uniform(-5, 5, size=3) if k else np.zeros(3)
        trans = np.array([0.0, 0.0, 0.0]); scale = 1.0                     # the reference's defaults (scene/cameras.py:18)
        w2v = ns["getWorld2View2"](R, T, trans, scale)                       # untransposed, column-vector convention
        env = {"torch": torch, "getWorld2View2": ns["getWorld2View2"], "R": R, "T": T, "trans": trans, "scale": scale}
        V = eval(exprs["world_view_transform"], env)                         # what raster_settings.viewmatrix holds
        class _Self: pass
        s = _Self(); s.world_view_transform = V
        cc = eval(exprs["camera_center"], {"self": s})
        Rs.append(R); Ts.append(T); Vs.append(V.numpy()); Cs.append(cc.numpy()); W2Vs.append(w2v)
    out.update(view_R=np.stack(Rs), view_T=np.stack(Ts), view_world_view_transform=np.stack(Vs).astype(np.float32),
               view_camera_center=np.stack(Cs).astype(np.float32), view_w2v_untransposed=np.stack(W2Vs).astype(np.float32),
               view_points=rng.uniform(-40, 40, size=(64, 3)).astype(np.float32))
    # ---- 2. covariance -------------------------------------------------------------------------------------------------
    ns2 = functions_of(os.path.join(REF, "utils", "general_utils.py"),
                       ["strip_lowerdiag", "strip_symmetric", "build_rotation", "build_scaling_rotation"], {"torch": torch})
    cov_fn = gaussian_model_cov_function(ns2)
    P = 256
    scales = np.exp(rng.normal(size=(P, 3)) * 0.7 - 2.0).astype(np.float32)
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = q.astype(np.float32)
    covs = {}
    with CudaToCpu():
        for mod in (1.0, 1.7):
            covs[mod] = cov_fn(torch.from_numpy(scales), mod, torch.from_numpy(q)).numpy()
        rot = ns2["build_rotation"](torch.from_numpy(q)).numpy()
    out.update(cov_scales=scales, cov_rotations=q, cov_mod1=covs[1.0].astype(np.float32), cov_mod17=covs[1.7].astype(np.float32),
               cov_rotation_matrices=rot.astype(np.float32))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conventions_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
