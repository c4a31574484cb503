"""Full-size check of the live-row gradient exchange: world 8 virtual ranks on cfg3 / cfg4, ship_live on against off -- the dense gradients a rank
ends with must be identical (shells) / equal up to the order of 2-3 addends (wedges).   python tools/check_live.py [cfg] [shell|wedge]"""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import lidargs_dist, lidargs_scenes as sc
from test_dist_gpu import ThreadComm, _run_rank
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
wedges = (sys.argv[2] if len(sys.argv) > 2 else "shell") == "wedge"
kind, P, H, W, seed = sc.BASELINE_CONFIGS[cfg]
scene = sc.make_scene(kind, P, H, seed)
grads = sc.upstream_grads(H, W, seed)
world = 8
out = {}
for live in ("0", "1"):
    os.environ["LIDARGS_SHIP_LIVE"] = live
    shared = ThreadComm.Shared(world); results = [None] * world
    th = [threading.Thread(target=_run_rank, args=(shared, r, scene, W, H, grads, "reduce_scatter", results, wedges)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=600)
    for r in results:
        if isinstance(r, Exception): raise r
    out[live] = results
for r in range(world):
    for k in ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations"):
        a, b = out["0"][r][k], out["1"][r][k]
        nz = int((np.abs(a).reshape(a.shape[0], -1).max(1) > 0).sum())
        d = float(np.abs(a - b).max()); m = float(np.abs(a).max())
        if k == "dL_dopacity" or d > 0:
            print(f"rank {r} {k}: non-zero rows {nz}, max |diff| {d:.3e} (max |value| {m:.3e})")
        assert d <= 1e-4 * m, (r, k, d, m)                  # (two runs of the backward differ by its atomics' order)
        assert np.array_equal(np.abs(a).reshape(a.shape[0], -1).max(1) > 0, np.abs(b).reshape(b.shape[0], -1).max(1) > 0), (r, k)
print("identical" if not wedges else "equal up to addend order")
