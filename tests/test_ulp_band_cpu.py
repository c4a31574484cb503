"""The error bar of "the reference": how far two CONFORMING evaluations of the reference source can be apart.

The reference rasterizer calls the float overloads of cos / sin / atan2 / tan / exp (R3/cr/forward.cu:589-591, :333-336, :361-362,
:604; R3/cr/backward.cu:659-661, :676) from CUDA libdevice (no -use_fast_math, R3/setup.py:29), whose documented maximum errors
are 1 / 1 / 2 / 4 / 2 ulp.  The oracle evaluates the same calls with the host libm (cos / sin correctly rounded).  Neither is "the"
answer; the reference's outputs are defined up to those last bits.  oracle/lidargs_oracle.c has a knob
(lgo_set_ulp_perturbation) that moves every such result by an integer number of ulps inside the documented bound; this test
runs the oracle against itself under the knob and

  * checks that with the knob off nothing changes (bit-identical), and that the knob really reaches all five functions;
  * MEASURES the band (relative difference with tests/util.py's metric) per output -- printed, and asserted against ceilings
    a little above the measured values so that the numbers quoted in DESIGN.md section 3 cannot silently rot:
      images (colour / depth / occupancy):  p99.9 <= 5e-5, no entry above 1e-4 except threshold flips (<= 2e-4 of the pixels)
      gradients: p99 up to 2e-4, p99.9 up to 1e-3, 0.1-2.5 % of the entries above 1e-4
    i.e. the north star's "within 1e-4" is attainable entry by entry for the images, and only statistically for the
    gradients -- by ANY implementation, the reference on another GPU generation included.
"""
import ctypes as C

import numpy as np
import pytest

import lidargs_scenes as sc
from util import GRAD_KEYS_SR, oracle_forward_backward

IMAGES = ("color", "depth", "occ")


def _run(scene, W, H, grads, mode, seed=0):
    from oracle import lgo
    L = lgo.lib()
    L.lgo_set_ulp_perturbation(C.c_int(mode), C.c_uint(seed))
    try:
        return oracle_forward_backward(scene, W, H, grads)
    finally:
        L.lgo_set_ulp_perturbation(C.c_int(0), C.c_uint(0))


def _err(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return np.abs(a - b) / (np.abs(b) + 1e-3 * np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("case", [("cfg1", "shell", 10_000, 16, 512, 1), ("dense64", "street", 60_000, 64, 1000, 4)], ids=lambda c: c[0])
def test_transcendental_rounding_band(case):
    name, kind, P, H, W, seed = case
    scene = sc.make_scene(kind, P, H, seed)
    grads = sc.upstream_grads(H, W, seed)
    base = _run(scene, W, H, grads, 0)
    again = _run(scene, W, H, grads, 0)
    for k in IMAGES + GRAD_KEYS_SR + ("radii",):
        assert np.array_equal(base[k], again[k]), k                      # knob off = the plain oracle, bit for bit
    worst = {}
    for mode, sd in ((1, 1), (1, 2), (2, 0), (3, 0)):
        o = _run(scene, W, H, grads, mode, sd)
        assert int((o["radii"] != base["radii"]).sum()) <= max(1, P // 10_000)
        assert not np.array_equal(o["color"], base["color"])              # the knob reaches the blend
        for k in IMAGES + GRAD_KEYS_SR:
            e = _err(o[k], base[k])
            st = dict(p99=float(np.quantile(e, 0.99)), p999=float(np.quantile(e, 0.999)), max=float(e.max()), over=float((e > 1e-4).mean()))
            w = worst.setdefault(k, dict(p99=0.0, p999=0.0, max=0.0, over=0.0))
            for q in w:
                w[q] = max(w[q], st[q])
    print(f"\n[ulp band] {name}: worst over 4 perturbations (random x2, all +amp, all -amp); metric of tests/util.py")
    for k in IMAGES + GRAD_KEYS_SR:
        w = worst[k]
        print(f"[ulp band] {name:8s} {k:14s} p99 {w['p99']:.2e}  p99.9 {w['p999']:.2e}  max {w['max']:.2e}  entries > 1e-4: {w['over']:.2e}")
    for k in IMAGES:
        assert worst[k]["p999"] <= 5e-5 and worst[k]["over"] <= 2e-4, (k, worst[k])
    for k in GRAD_KEYS_SR:
        assert worst[k]["p99"] <= 4e-4 and worst[k]["p999"] <= 2e-3, (k, worst[k])
    # the point of the exercise: the gradients' band is NOT inside 1e-4 entry by entry
    assert max(worst[k]["over"] for k in ("dL_dmeans3D", "dL_dopacity")) > 1e-3


def test_each_function_is_reached():
    """One function at a time cannot be switched here, but the radii depend only on atan2f / tanf and the images only on cos / sin /
    exp: mode 2 (every result +amp ulp) must move the projected columns (atan2f) and the blend (exp, cos, sin)."""
    scene = sc.make_scene("shell", 4000, 16, 9)
    base = _run(scene, 512, 16, None, 0)
    up = _run(scene, 512, 16, None, 2)
    m0 = base["fwd"].array("means2D"); m1 = up["fwd"].array("means2D")
    assert m0.shape == m1.shape and not np.array_equal(m0, m1)            # atan2f
    assert np.abs(m0 - m1).max() < 1e-2                                   # ... by ulps, not pixels
    assert not np.array_equal(base["occ"], up["occ"])                     # expf / cosf / sinf
