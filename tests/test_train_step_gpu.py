"""End-to-end chain of the three fused components on the GPU -- anchor decode (f1) -> rasterizer (a1-a16) -> image loss (f2) ->
backward to the anchor features, offsets and MLP weights -- against the same chain of CPU oracles."""
import types

import numpy as np
import pytest

import lidargs_scenes as sc
from oracle import lgo, lidar_loss as oloss, neural_gaussians as ong
from test_neural_gaussians_cpu import PARAM_KEYS
from test_neural_gaussians_gpu import build_pc, random_case
from util import make_settings, parity

pytestmark = pytest.mark.gpu
# (no outlier-budget override any more: tests/util.py's soft / flip classes apply as everywhere else)


def test_decode_rasterize_loss_chain_matches_oracles(hip_lib_built):
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from lidar_loss import image_loss
    from neural_gaussians import generate_neural_gaussians
    N, k, H, W, seed = 4000, 6, 16, 512, 21
    p, cam, vis, rng = random_case(N, k, seed)
    # put the anchors on a shell around the sensor so that they are in view, small offsets / scalings so that they look like a scene
    d = rng.normal(size=(N, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True); d[:, 2] = np.abs(d[:, 2]) * -0.25 * np.sign(1.0)
    p["anchor"] = (d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(4, 40, size=(N, 1))).astype(np.float32)
    p["offset"] = (0.2 * rng.normal(size=(N, k, 3))).astype(np.float32)
    p["scaling"] = np.exp(rng.normal(size=(N, 6)) * 0.3 - 1.6).astype(np.float32)
    cam = np.zeros(3, np.float32)
    scene = sc.make_scene("shell", 10, H, seed)            # only for the sensor model: beams, identity view matrix, background
    gt = np.stack([(rng.random((H, W)) > 0.2).astype(np.float32), rng.random((H, W), dtype=np.float32),
                   (rng.random((H, W)) * 40).astype(np.float32)])
    lam = 0.2

    # ---- oracle chain
    f = ong.forward(p, cam, vis)
    fw = lgo.forward(f["xyz"], f["color"], f["opacity"], f["scaling"], f["rot"], scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"])
    lo = oloss.forward_backward(fw.color, fw.depth, gt, lam)
    gb = lgo.backward(fw, lo["g_image"], lo["g_depth"], np.zeros((1, H, W), np.float32))
    reg, g_reg = oloss.scaling_reg(f["scaling"])                        # train.py:174: the per-Gaussian term of the loss
    g = ong.backward(p, f, gb["dL_dmeans3D"], gb["dL_dcolors"], gb["dL_dopacity"], gb["dL_dscales"] + g_reg, gb["dL_drotations"])

    # ---- product chain
    pc = build_pc(p)
    camera = types.SimpleNamespace(camera_center=torch.from_numpy(cam).cuda(), uid=0)
    xyz, color, opacity, scaling, rot, _no, _mask = generate_neural_gaussians(camera, pc, torch.from_numpy(vis).cuda(), is_training=True)
    st = {k_: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k_, v in scene.items()}
    rast = GaussianRasterizer(make_settings(st, W, H))
    means2D = torch.zeros((xyz.shape[0], 4), device="cuda", requires_grad=True)
    image, depth, occ, radii = rast(means3D=xyz, means2D=means2D, opacities=opacity, colors_precomp=color, scales=scaling, rotations=rot)
    terms = image_loss(image, depth, torch.from_numpy(gt).cuda(), lam, scaling=scaling)      # the reference's whole `loss`, scaling_reg included
    terms["loss"].backward()

    assert xyz.shape[0] == f["xyz"].shape[0] and int((radii.cpu().numpy() != fw.radii).sum()) <= 1
    assert (fw.radii > 0).sum() > 500                                   # the scene is really rendered
    parity("image", image.detach().cpu().numpy(), fw.color); parity("depth", depth.detach().cpu().numpy(), fw.depth)
    assert abs(float(terms["loss"]) - (lo["loss"] + reg)) <= 1e-4 * abs(lo["loss"] + reg)
    assert abs(float(terms["scaling_reg"]) - reg) <= 1e-4 * abs(reg)
    parity("d anchor_feat", pc._anchor_feat.grad.cpu().numpy(), g["anchor_feat"])
    parity("d anchor", pc._anchor.grad.cpu().numpy(), g["anchor"])
    parity("d offset", pc._offset.grad.cpu().numpy(), g["offset"])
    parity("d scaling", pc.get_scaling.grad.cpu().numpy(), g["scaling"])
    for name in ong.MLPS:
        seq = getattr(pc, "mlp_" + name)
        parity(f"d {name}_W1", seq[0].weight.grad.cpu().numpy(), g[name + "_W1"], rtol=5e-4)
        parity(f"d {name}_W2", seq[2].weight.grad.cpu().numpy(), g[name + "_W2"], rtol=5e-4)
