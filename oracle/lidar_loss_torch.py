"""The per-frame image loss as the reference writes it -- a chain of framework ops (train.py:150-203, utils/loss_utils.py:20-64)
-- restated with torch so that it runs on any device.  MEASUREMENT INFRASTRUCTURE ONLY ("what the reference's graph costs on
this GPU" in bench.py --workload loss); the product never imports it."""
from math import exp

import torch
import torch.nn.functional as F


def _window(channel, like):
    g = torch.tensor([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, 11, 11).contiguous().to(like)


def ssim(img1, img2):
    w = _window(img1.size(-3), img1)
    c = img1.size(-3)
    mu1, mu2 = F.conv2d(img1, w, padding=5, groups=c), F.conv2d(img2, w, padding=5, groups=c)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, w, padding=5, groups=c) - mu1_sq
    s2 = F.conv2d(img2 * img2, w, padding=5, groups=c) - mu2_sq
    s12 = F.conv2d(img1 * img2, w, padding=5, groups=c) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))).mean()


def image_loss(image, depth, gt_image, lambda_dssim):
    l1 = lambda a, b: torch.abs(a - b).mean()
    ray_drop = gt_image[0:1]
    gt_intensity, gt_depth = gt_image[1:2] * ray_drop, gt_image[2:3] * ray_drop
    render_intensity, render_raydrop = image[0:1] * ray_drop, image[1:2]
    depth = depth * ray_drop
    raydrop_loss = 10 * torch.nn.functional.mse_loss(render_raydrop, ray_drop)
    Ll1, depth_loss = l1(render_intensity, gt_intensity), l1(depth, gt_depth)
    ssim_loss = 1.0 - ssim(render_intensity, gt_intensity)
    pred_grad_x = torch.abs(depth[:, :, :-1] - depth[:, :, 1:])
    gt_grad_x = torch.abs(gt_depth[:, :, :-1] - gt_depth[:, :, 1:])
    mask_dx = ray_drop[:, :, :-1] * torch.where(gt_grad_x < 0.01, 1, 0)
    grad_loss = l1(pred_grad_x * mask_dx, gt_grad_x * mask_dx)
    return depth_loss + (1.0 - lambda_dssim) * Ll1 + lambda_dssim * ssim_loss + raydrop_loss + grad_loss
