"""Where HIP differs from the oracle by more than 1e-4 in the semi-transparent regime (cfg3 with opacities x 0.1, one azimuth wedge) and on the
listed sweep scenes: is it the oracle's summation (exact float64 sums of the oracle's own float32 terms), or the reference's own band
(seven conforming evaluations: ulp-perturbed cos / sin / atan2 / tan / exp, reversed atomics order, contracted multiply-adds)?
    python tools/thin_residue.py [thin] [seed ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import lidargs_scenes as sc
import util
from util import (GRAD_KEYS_SR, GRAD_KEYS_SURFEL, envelope_residue, hip_forward_backward, hip_surfel_forward_backward, oracle_backward_exact_sums,
                  oracle_envelope, oracle_forward_backward, oracle_surfel_envelope, oracle_surfel_forward_backward)


def report(tag, hip, base, lo, hi, r64, keys, rows=None, inside=None, r64t=None):
    for k in keys:
        h = hip[k] if rows is None else hip[k][rows]
        hh = {k: h if inside is None else h[inside]}
        sel = (lambda a: a) if inside is None else (lambda a: a[inside])
        st = envelope_residue(hh, {k: sel(base[k])}, {k: sel(lo[k])}, {k: sel(hi[k])}, k)
        r = np.asarray(sel(r64[k]), np.float64); b = np.asarray(sel(base[k]), np.float64); x = np.asarray(hh[k], np.float64)
        den = np.abs(r) + 1e-3 * np.abs(r).max() + 1e-30
        print(f"[{tag}] {k:14s} n={st['n']:7d} hip>1e-4 vs oracle: {st['hip_over']:5d} (in the band: {st['hip_over_where_oracle_moves_half']:5d}, band itself >1e-4: {st['oracle_band_over']:5d}, "
              f"worst outside {st['worst_outside_anywhere']:.2f} widths) | vs exact sums: hip {int((np.abs(x - r) / den > 1e-4).sum()):5d}, oracle {int((np.abs(b - r) / den > 1e-4).sum()):5d}"
              + ("" if r64t is None else f" | vs exact sums + float64 T chain: hip {int((np.abs(x - np.asarray(sel(r64t[k]), np.float64)) / den > 1e-4).sum()):5d}, oracle {int((np.abs(b - np.asarray(sel(r64t[k]), np.float64)) / den > 1e-4).sum()):5d}"), flush=True)


args = sys.argv[1:] or ["thin"]
for a in args:
    if a == "thin":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_fullsize_gpu import _wedge_subset
        kind, P, H, W, seed = sc.BASELINE_CONFIGS["cfg3"]
        scene = sc.make_scene(kind, P, H, seed, opacity_scale=0.1)
        grads = sc.upstream_grads(H, W, seed)
        hip = hip_forward_backward(scene, W, H, grads)
        c0 = (W // 2 - 96) // 16 * 16; c1 = c0 + 192
        keep = _wedge_subset(scene, W, hip["radii"], c0, c1)
        sub = dict(scene)
        for k in ("means3D", "scales", "rotations", "opacities", "colors"):
            sub[k] = np.ascontiguousarray(scene[k][keep])
        base, lo, hi = oracle_envelope(sub, W, H, grads, {}, GRAD_KEYS_SR)
        r64 = oracle_backward_exact_sums(base, grads)
        rows = np.nonzero(keep)[0]
        m2 = base["fwd"].array("means2D").reshape(-1, 2); rx = base["fwd"].array("radii_xy").reshape(-1, 2)[:, 0].astype(np.float64)
        inside = (base["radii"] > 0) & (hip["radii"][rows] > 0) & (np.floor((m2[:, 0] - rx) / 16.0) * 16 >= c0) & (np.floor((m2[:, 0] + rx + 15.0) / 16.0) * 16 <= c1)
        report("cfg3_thin wedge", hip, base, lo, hi, r64, GRAD_KEYS_SR, rows, inside)
    else:
        c = sc.sweep_case_any(int(a), mid=False)
        scene, W, H, grads, kw = c["scene"], c["W"], c["H"], c["grads"], c["kw"]
        fast = os.environ.get("NO_ENVELOPE") == "1"                     # the plain oracle only (no seven-run band)
        if c["surfel"]:
            hip = hip_surfel_forward_backward(scene, W, H, grads, **kw)
            keys = GRAD_KEYS_SURFEL
            if fast:
                base = oracle_surfel_forward_backward(scene, W, H, grads, **kw); lo = hi = {k: np.asarray(base[k], np.float64) for k in keys}
            else:
                base, lo, hi = oracle_surfel_envelope(scene, W, H, grads, kw, GRAD_KEYS_SURFEL)
        else:
            hip = hip_forward_backward(scene, W, H, grads, cov3D_precomp=c["cov"], **kw)
            keys = GRAD_KEYS_SR if c["cov"] is None else ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dcov3D")
            if fast:
                base = oracle_forward_backward(scene, W, H, grads, cov3D_precomp=c["cov"], **kw); lo = hi = {k: np.asarray(base[k], np.float64) for k in keys}
            else:
                base, lo, hi = oracle_envelope(scene, W, H, grads, dict(kw, cov3D_precomp=c["cov"]), keys)
        r64 = oracle_backward_exact_sums(base, grads, surfel=c["surfel"])
        r64t = oracle_backward_exact_sums(base, grads, surfel=c["surfel"], mode=2)
        print(c["desc"])
        report(a, hip, base, lo, hi, r64, keys, r64t=r64t)
