"""Generates tests/golden/lidar_loss_golden.npz by EXECUTING the reference's own loss code on CPU torch:
the statements `gt_image = ...` through `loss = ...` of training() (/root/reference/train.py:150-203) and the functions
l1_loss / gaussian / create_window / ssim / _ssim of /root/reference/utils/loss_utils.py:20-64, followed by loss.backward().

The source is read from the reference checkout at run time (nothing of it is stored here).  The statements are taken out of
the parsed function body by position (first assignment to `gt_image` ... last assignment to `loss`); the only adaptation is the
input object: `viewpoint_cam.original_image` is a tensor wrapper whose `.cuda()` returns the CPU tensor (no GPU in this image).

    python tests/golden/make_loss_golden.py
"""
import ast
import os
import types
from math import exp

import numpy as np
import torch
import torch.nn.functional as F
from torch.autograd import Variable

TRAIN = "/root/reference/train.py"
LOSS_UTILS = "/root/reference/utils/loss_utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lidar_loss_golden.npz")


def loss_functions():
    tree = ast.parse(open(LOSS_UTILS).read())
    want = {"l1_loss", "gaussian", "create_window", "ssim", "_ssim"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"torch": torch, "F": F, "Variable": Variable, "exp": exp}
    exec(compile(ast.fix_missing_locations(mod), LOSS_UTILS, "exec"), ns)
    return ns


def loss_statements():
    tree = ast.parse(open(TRAIN).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "training")
    loop = next(n for n in fn.body if isinstance(n, ast.For) and getattr(n.target, "id", "") == "iteration")
    names = [(i, s.targets[0].id) for i, s in enumerate(loop.body) if isinstance(s, ast.Assign) and isinstance(s.targets[0], ast.Name)]
    first = next(i for i, n in names if n == "gt_image")
    last = max(i for i, n in names if n == "loss")
    return ast.Module(body=loop.body[first:last + 1], type_ignores=[])


class HostImage:
    def __init__(self, t): self.t = t
    def cuda(self): return self.t


def run(tag, H, W, M, seed, out):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(2, H, W, generator=g).requires_grad_(True)
    depth = (torch.rand(1, H, W, generator=g) * 60.0).requires_grad_(True)
    ray_drop = (torch.rand(1, H, W, generator=g) > 0.25).float()
    gt_depth = torch.rand(1, H, W, generator=g) * 60.0
    # smooth runs so that the depth-gradient mask (|dx| < 0.01) selects something
    gt_depth[:, :, ::3] = gt_depth[:, :, 1::3][:, :, :gt_depth[:, :, ::3].shape[2]] + 0.004 if W % 3 == 0 else gt_depth[:, :, ::3]
    gt = torch.cat([ray_drop, torch.rand(1, H, W, generator=g), gt_depth], 0)
    scaling = (torch.rand(M, 3, generator=g) * 0.3 + 0.01).requires_grad_(True)
    ns = loss_functions()
    ns.update(torch=torch, image=image, depth=depth, scaling=scaling, viewpoint_cam=types.SimpleNamespace(original_image=HostImage(gt)),
              gaussians=types.SimpleNamespace(color_channel=2), opt=types.SimpleNamespace(lambda_dssim=0.2))
    exec(compile(ast.fix_missing_locations(loss_statements()), TRAIN, "exec"), ns)
    ns["loss"].backward()
    npy = lambda t: t.detach().numpy().astype(np.float32)
    out.update({f"{tag}_image": npy(image), f"{tag}_depth": npy(depth), f"{tag}_gt": npy(gt), f"{tag}_scaling": npy(scaling),
                f"{tag}_lambda_dssim": np.float32(0.2),
                f"{tag}_loss": npy(ns["loss"]), f"{tag}_Ll1": npy(ns["Ll1"]), f"{tag}_depth_loss": npy(ns["depth_loss"]),
                f"{tag}_ssim_loss": npy(ns["ssim_loss"]), f"{tag}_raydrop_loss": npy(ns["raydrop_loss"]), f"{tag}_grad_loss": npy(ns["grad_loss"]),
                f"{tag}_scaling_reg": npy(ns["scaling_reg"]),
                f"{tag}_g_image": npy(image.grad), f"{tag}_g_depth": npy(depth.grad), f"{tag}_g_scaling": npy(scaling.grad)})


if __name__ == "__main__":
    out = {}
    run("a", 16, 96, 50, 1, out)
    run("b", 64, 265, 80, 2, out)       # the headline height, a tenth of its width; W not a multiple of anything convenient
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
