"""CPU restatement (numpy) of LiDAR-GS's per-frame image loss and its gradient:
/root/reference/train.py:150-203 with l1_loss / ssim of /root/reference/utils/loss_utils.py:20-64.

TEST INFRASTRUCTURE ONLY.  PARITY STATUS: pinned -- tests/golden/lidar_loss_golden.npz holds inputs, the loss terms and the
autograd gradients produced by EXECUTING those reference statements on CPU torch (tests/golden/make_loss_golden.py).

loss_image = depth_loss + (1-l) Ll1 + l (1 - ssim) + 10 mse(raydrop) + grad_loss      (scaling_reg is per-Gaussian, not here)
"""
from math import exp

import numpy as np

WINDOW = 11
C1, C2 = 0.01 ** 2, 0.03 ** 2           # loss_utils.py:54-55
GRAD_CLIP_X = 0.01                      # train.py:194
RAYDROP_WEIGHT = 10.0                   # train.py:165


def window_1d():
    g = np.array([exp(-(x - WINDOW // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(WINDOW)], np.float32)     # loss_utils.py:27-29
    return (g / g.sum(dtype=np.float32)).astype(np.float32)


def _conv(img, w2d):
    """F.conv2d(img, window, padding=5) for one channel: zero-padded cross-correlation."""
    H, W = img.shape
    pad = WINDOW // 2
    p = np.zeros((H + 2 * pad, W + 2 * pad), np.float64)
    p[pad:pad + H, pad:pad + W] = img
    out = np.zeros((H, W), np.float64)
    for dy in range(WINDOW):
        for dx in range(WINDOW):
            out += w2d[dy, dx] * p[dy:dy + H, dx:dx + W]
    return out


def forward_backward(image, depth, gt, lambda_dssim):
    """image [2,H,W] (intensity, ray-drop), depth [1,H,W], gt [3,H,W] (ray-drop mask, intensity, depth).
    Returns dict(loss, Ll1, depth_loss, ssim_loss, raydrop_loss, grad_loss, g_image [2,H,W], g_depth [1,H,W])."""
    f = np.float64
    image, depth, gt = image.astype(f), depth.astype(f), gt.astype(f)
    H, W = image.shape[1:]
    N = H * W
    lam = float(lambda_dssim)
    rd = gt[0]
    gi, gd = gt[1] * rd, gt[2] * rd                              # :152-153
    X = image[0] * rd                                            # :161
    dm = depth[0] * rd                                           # :162
    rr = image[1]
    raydrop_loss = RAYDROP_WEIGHT * ((rr - rd) ** 2).mean()      # :163-165
    Ll1 = np.abs(X - gi).mean()                                  # :171
    depth_loss = np.abs(dm - gd).mean()                          # :172
    w1 = window_1d()
    w2d = (w1[:, None].astype(np.float32) @ w1[None, :].astype(np.float32)).astype(np.float32).astype(f)       # loss_utils.py:32-33
    mu1, mu2 = _conv(X, w2d), _conv(gi, w2d)
    p11, p22, p12 = _conv(X * X, w2d), _conv(gi * gi, w2d), _conv(X * gi, w2d)
    s11, s22, s12 = p11 - mu1 * mu1, p22 - mu2 * mu2, p12 - mu1 * mu2
    A1, A2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    B1, B2 = mu1 * mu1 + mu2 * mu2 + C1, s11 + s22 + C2
    S = (A1 * A2) / (B1 * B2)                                    # loss_utils.py:57
    ssim_loss = 1.0 - S.mean()                                   # :173
    pg = np.abs(dm[:, :-1] - dm[:, 1:])                          # :190
    gg = np.abs(gd[:, :-1] - gd[:, 1:])                          # :193
    mask = rd[:, :-1] * (gg < GRAD_CLIP_X)                       # :195-199
    grad_loss = np.abs(pg * mask - gg * mask).mean()             # :201
    loss = depth_loss + (1.0 - lam) * Ll1 + lam * ssim_loss + raydrop_loss + grad_loss     # :203-205 minus scaling_reg
    # ---- gradients
    g_image = np.zeros((2, H, W), f)
    g_depth = np.zeros((1, H, W), f)
    g_image[1] = RAYDROP_WEIGHT * 2.0 * (rr - rd) / N
    gX = (1.0 - lam) * np.sign(X - gi) / N
    dS = -lam / N                                                # d loss / d S per pixel
    dS_ds12 = 2 * A1 / (B1 * B2)
    dS_ds11 = -(A1 * A2) / (B1 * B2 * B2)
    dS_dm1 = (2 * mu2 * A2) / (B1 * B2) - (A1 * A2) * (2 * mu1) / (B1 * B1 * B2) - dS_ds12 * mu2 - dS_ds11 * 2 * mu1
    gX = gX + _conv(dS * dS_dm1, w2d) + 2 * X * _conv(dS * dS_ds11, w2d) + gi * _conv(dS * dS_ds12, w2d)
    g_image[0] = gX * rd
    gdm = np.sign(dm - gd) / N
    gp = mask * np.sign(pg * mask - gg * mask) / (H * (W - 1))
    sgn = np.sign(dm[:, :-1] - dm[:, 1:])
    gdm[:, :-1] += gp * sgn
    gdm[:, 1:] -= gp * sgn
    g_depth[0] = gdm * rd
    return dict(loss=loss, Ll1=Ll1, depth_loss=depth_loss, ssim_loss=ssim_loss, raydrop_loss=raydrop_loss, grad_loss=grad_loss,
                g_image=g_image.astype(np.float32), g_depth=g_depth.astype(np.float32))


def scaling_reg(scaling, weight=0.01):
    """train.py:174  scaling_reg = 0.01 * scaling.prod(dim=1).mean()  and its gradient w.r.t. scaling [M, 3] (f64 accumulation)."""
    s = np.asarray(scaling, np.float64)
    M = s.shape[0]
    value = weight * np.prod(s, axis=1).mean() if M else float("nan")
    g = np.stack([s[:, 1] * s[:, 2], s[:, 0] * s[:, 2], s[:, 0] * s[:, 1]], 1) * (weight / max(M, 1))
    return float(value), g.astype(np.float32)
