"""Anchor growing far above the training size: N anchors x k offsets with EVERY masked offset a candidate (threshold 0), native call against the numpy oracle.
    python tools/anchor_stress.py [N_millions] [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import lidargs_scenes as sc
import anchor_growing as ag
from oracle import anchor_growing as oag
N = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 3_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = sc.anchor_scene(N, k, 31)
N = c["N"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
act = np.exp(c["scaling"]).astype(np.float32)
args = (t(c["anchor"]), t(c["offset"]), t(act), t(c["feat"]))
for size in (0.16, 0.01):
    torch.cuda.synchronize(); t0 = time.time()
    ga, gf, counts = ag.grow_level(*args, t(c["grads"]), t(c["offset_mask"]), None, 0.0, 0.5, size)
    torch.cuda.synchronize(); dt = time.time() - t0
    cand = oag.candidate_mask(c["grads"], c["offset_mask"], None, 0.0, 0.5, N * k)
    t0 = time.time()
    ra, rf, n_c, n_v = oag.grow_level(c["anchor"], c["offset"], act, c["feat"], cand, size, exact_division=False)
    print(dict(N=N, k=k, voxel_edge=size, counts=counts, native_ms=round(dt * 1e3, 2), oracle_s=round(time.time() - t0, 1),
               equal=bool(counts == (n_c, n_v, ra.shape[0]) and np.array_equal(ga.cpu().numpy(), ra) and np.array_equal(gf.cpu().numpy(), rf)),
               mem_GB=round(torch.cuda.max_memory_allocated() / 2**30, 2)), flush=True)
