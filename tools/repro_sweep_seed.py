"""One scene of tools/parity_sweep.py by seed, with the image planes compared entry by entry:  python tools/repro_sweep_seed.py 5017"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import lidargs_scenes as sc
from util import hip_forward_backward, oracle_forward_backward
seed = int(sys.argv[1]); mid = (sys.argv[2] == "mid") if len(sys.argv) > 2 else seed >= 100000      # python tools/repro_sweep_seed.py <seed> [small|mid]
scene, W, H, grads, kw, desc = sc.sweep_case(seed, mid)
kind, P, beams = desc['kind'], desc['P'], desc['beams']
hip = hip_forward_backward(scene, W, H, grads, **kw)
ref = oracle_forward_backward(scene, W, H, grads, **kw)
from diff_lidargs_rasterization import _C
_C.counters_enable(True)      # diagnostics tool: every forward ends with the counting launches of last_counters()
print(dict(seed=seed, kind=kind, P=P, H=H, W=W, beams=beams, **kw), _C.last_counters())
for k in ("color", "depth", "occ"):
    d = np.abs(hip[k] - ref[k]); bad = np.argwhere(d > 1e-3 * (np.abs(ref[k]) + 1e-3 * np.abs(ref[k]).max()))
    print(k, "bad", len(bad), "of", d.size, "max", d.max(), "first", [tuple(int(x) for x in b) for b in bad[:6]], [float(hip[k][tuple(b)]) for b in bad[:3]], [float(ref[k][tuple(b)]) for b in bad[:3]],
          "rows of the bad pixels:", sorted(set(int(b[1]) for b in bad)) if len(bad) else [])
print("radii mismatches", int((hip["radii"] != ref["radii"]).sum()), "visible", int((ref["radii"] > 0).sum()))
from util import GRAD_KEYS_SR
for k in GRAD_KEYS_SR:
    r = ref[k].astype(np.float64); h = hip[k].astype(np.float64)
    err = np.abs(h - r) / (np.abs(r) + 1e-3 * np.abs(r).max() + 1e-30)
    rows = np.unique(np.argwhere(err > 1e-4)[:, 0])
    print(f"{k:14s} n={r.size:8d} soft={(int(((err > 1e-4) & (err <= 1e-3)).sum())):5d} flips={int((err > 1e-3).sum()):4d} max={err.max():.2e} rows affected={len(rows)}")
