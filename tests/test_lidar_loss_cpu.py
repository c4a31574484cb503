"""Pins oracle/lidar_loss.py (row f2: the per-frame image loss and its gradient) against golden vectors produced by executing
the reference's own loss statements (train.py:150-203, utils/loss_utils.py) and loss.backward() on CPU torch."""
import os

import numpy as np
import pytest

from oracle import lidar_loss

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lidar_loss_golden.npz")


def load(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}


def rel(name, got, ref, rtol=1e-4, floor=1e-3):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref) / (np.abs(ref) + floor * max(np.abs(ref).max(), 1e-30))
    assert err.max() <= rtol, f"{name}: {err.max():.3e}"


@pytest.mark.parametrize("tag", ["a", "b"])
def test_loss_and_gradients_match_reference_execution(tag):
    c = load(tag)
    r = lidar_loss.forward_backward(c["image"], c["depth"], c["gt"], float(c["lambda_dssim"]))
    for k in ("Ll1", "depth_loss", "ssim_loss", "raydrop_loss", "grad_loss"):
        rel(k, r[k], c[k], rtol=2e-5)
    reg, g_scaling = lidar_loss.scaling_reg(c["scaling"])                 # train.py:174, the per-Gaussian term of the reference's `loss`
    rel("scaling_reg", reg, c["scaling_reg"], rtol=2e-5)
    rel("loss", r["loss"] + reg, c["loss"], rtol=2e-5)
    rel("g_scaling", g_scaling, c["g_scaling"])
    rel("g_image", r["g_image"], c["g_image"])
    rel("g_depth", r["g_depth"], c["g_depth"])


def test_window_matches_reference_normalisation():
    w = lidar_loss.window_1d()
    assert w.shape == (11,) and abs(float(w.sum()) - 1.0) < 1e-6 and np.argmax(w) == 5 and np.allclose(w, w[::-1])
