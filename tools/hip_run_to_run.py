"""Run-to-run spread of the HIP gradients (float atomics in scheduling order) on sweep scenes:  python tools/hip_run_to_run.py seed ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import lidargs_scenes as sc
from util import GRAD_KEYS_SR, GRAD_KEYS_SURFEL, hip_forward_backward, hip_surfel_forward_backward
for a in sys.argv[1:]:
    c = sc.sweep_case_any(int(a), mid=False)
    scene, W, H, grads, kw = c["scene"], c["W"], c["H"], c["grads"], c["kw"]
    keys = GRAD_KEYS_SURFEL if c["surfel"] else (GRAD_KEYS_SR if c["cov"] is None else ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dcov3D"))
    runs = [hip_surfel_forward_backward(scene, W, H, grads, **kw) if c["surfel"] else hip_forward_backward(scene, W, H, grads, cov3D_precomp=c["cov"], **kw) for _ in range(4)]
    for k in keys:
        x = np.stack([np.asarray(r[k], np.float64) for r in runs])
        den = np.abs(x[0]) + 1e-3 * np.abs(x[0]).max() + 1e-30
        spread = (x.max(0) - x.min(0)) / den
        print(f"[{a}] {k:14s} run-to-run spread over 4 runs: max {spread.max():.2e}, entries > 1e-4: {int((spread > 1e-4).sum())}, > 1e-5: {int((spread > 1e-5).sum())} of {spread.size}")
