"""GPU parity of the fused image loss: `lidar_loss.image_loss` (C ABI lidargs_image_loss) against the
golden vectors from executing the reference's loss statements + autograd, and against the numpy oracle at the headline image size."""
import numpy as np
import pytest

from oracle import lidar_loss as oracle_loss
from test_lidar_loss_cpu import load
from util import parity

pytestmark = pytest.mark.gpu
TERMS = ("Ll1", "depth_loss", "ssim_loss", "raydrop_loss", "grad_loss")


def run_hip(image, depth, gt, lam, scaling=None):
    import torch
    from lidar_loss import image_loss
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    img, dep = t(image).requires_grad_(True), t(depth).requires_grad_(True)
    sc = t(scaling).requires_grad_(True) if scaling is not None else None
    terms = image_loss(img, dep, t(gt), lam, scaling=sc)           # with scaling: the reference's whole `loss` (train.py:174-175 included)
    loss = terms["loss"]
    loss.backward()
    out = {k: float(terms[k]) for k in TERMS + ("scaling_reg",)}
    out.update(loss=float(loss), g_image=img.grad.cpu().numpy(), g_depth=dep.grad.cpu().numpy())
    if sc is not None:
        out["g_scaling"] = sc.grad.cpu().numpy()
    return out


@pytest.mark.parametrize("tag", ["a", "b"])
def test_loss_matches_reference_golden(tag, hip_lib_built):
    c = load(tag)
    r = run_hip(c["image"], c["depth"], c["gt"], float(c["lambda_dssim"]), c["scaling"])
    for k in TERMS + ("loss", "scaling_reg"):
        assert abs(r[k] - float(c[k])) <= 2e-5 * abs(float(c[k])) + 1e-7, (k, r[k], float(c[k]))
    parity("g_image", r["g_image"], c["g_image"])
    parity("g_depth", r["g_depth"], c["g_depth"])
    parity("g_scaling", r["g_scaling"], c["g_scaling"])            # the native scaling_reg's gradient against the reference's autograd


def test_loss_matches_oracle_at_headline_size(hip_lib_built):
    rng = np.random.default_rng(3)
    H, W = 64, 2650
    image = rng.random((2, H, W), dtype=np.float32)
    depth = (rng.random((1, H, W), dtype=np.float32) * 70).astype(np.float32)
    gt = np.stack([(rng.random((H, W)) > 0.2).astype(np.float32), rng.random((H, W), dtype=np.float32),
                   np.cumsum(rng.normal(scale=0.004, size=(H, W)), axis=1).astype(np.float32) + 20.0])     # smooth rows: the |dx| < 0.01 mask is active
    ref = oracle_loss.forward_backward(image, depth, gt, 0.2)
    r = run_hip(image, depth, gt, 0.2)
    for k in TERMS + ("loss",):
        assert abs(r[k] - ref[k]) <= 2e-5 * abs(ref[k]) + 1e-7, (k, r[k], ref[k])
    parity("g_image", r["g_image"], ref["g_image"])
    parity("g_depth", r["g_depth"], ref["g_depth"])


def test_scaling_reg_sizes_and_a_scaled_upstream_gradient(hip_lib_built):
    """lidargs_scaling_reg at the sizes a decode hands over (0.9 M rows: several thousand block sums), a row count off every block
    size, one row; and the stored gradients scaled by an upstream factor (loss * 3).backward()."""
    import torch
    from lidar_loss import image_loss
    rng = np.random.default_rng(9)
    H, W = 8, 40
    image, depth = rng.random((2, H, W), dtype=np.float32), rng.random((1, H, W), dtype=np.float32)
    gt = rng.random((3, H, W), dtype=np.float32); gt[0] = gt[0] > 0.3
    base = oracle_loss.forward_backward(image, depth, gt, 0.2)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for M in (1, 255, 257, 900_001):
        scaling = np.exp(rng.normal(size=(M, 3)) * 0.5 - 1.5).astype(np.float32)
        reg, g_ref = oracle_loss.scaling_reg(scaling)
        sc, img, dep = t(scaling).requires_grad_(True), t(image).requires_grad_(True), t(depth).requires_grad_(True)
        terms = image_loss(img, dep, t(gt), 0.2, scaling=sc)
        (terms["loss"] * 3.0).backward()
        assert abs(float(terms["scaling_reg"]) - reg) <= 2e-5 * abs(reg), (M, float(terms["scaling_reg"]), reg)
        assert abs(float(terms["loss"]) - (base["loss"] + reg)) <= 2e-5 * abs(base["loss"] + reg)
        parity(f"g_scaling[M={M}]", sc.grad.cpu().numpy(), 3.0 * g_ref)
        parity("g_image x3", img.grad.cpu().numpy(), 3.0 * base["g_image"])
    with pytest.raises(RuntimeError, match="expected scaling"):
        image_loss(t(image), t(depth), t(gt), 0.2, scaling=torch.zeros(4, 2).cuda())


def test_loss_edge_shapes_and_errors(hip_lib_built):
    import torch
    from lidar_loss import image_loss
    rng = np.random.default_rng(4)
    for H, W in ((1, 2), (3, 7), (16, 257)):                      # smaller than the 11-tap window, odd sizes
        image, depth = rng.random((2, H, W), dtype=np.float32), rng.random((1, H, W), dtype=np.float32)
        gt = rng.random((3, H, W), dtype=np.float32); gt[0] = gt[0] > 0.3
        ref = oracle_loss.forward_backward(image, depth, gt, 0.2)
        r = run_hip(image, depth, gt, 0.2)
        assert abs(r["loss"] - ref["loss"]) <= 2e-5 * abs(ref["loss"]) + 1e-7
        parity("g_image", r["g_image"], ref["g_image"]); parity("g_depth", r["g_depth"], ref["g_depth"])
    with pytest.raises(RuntimeError, match="HIP device"):
        image_loss(torch.zeros(2, 4, 8), torch.zeros(1, 4, 8), torch.zeros(3, 4, 8))
    with pytest.raises(RuntimeError, match="expected image"):
        image_loss(torch.zeros(3, 4, 8).cuda(), torch.zeros(1, 4, 8).cuda(), torch.zeros(3, 4, 8).cuda())
