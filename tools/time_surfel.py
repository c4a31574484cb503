"""Time the surfel variant (forward, backward) on a BASELINE-sized scene.  usage: time_surfel.py [P] [H] [W] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import lidargs_scenes as sc
from diff_lidargs_surfel_rasterization import _C

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2650
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
s = sc.make_scene("street", P, H, 5)
s["scales"] = np.ascontiguousarray(s["scales"][:, :2])
t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in s.items()}
e = torch.empty(0, device="cuda")
eye, z3 = torch.eye(4).cuda(), torch.zeros(3).cuda()
gc = torch.randn(2, H, W, device="cuda"); go = torch.randn(7, H, W, device="cuda")


def fwd():
    return _C.rasterize_gaussians(t["bg"], t["means3D"], t["colors"], t["opacities"], t["scales"], t["rotations"], 1.0, e, t["viewmatrix"],
                                  eye, t["beams"], H, W, e, 1, z3, False, 80, 0, False)


def bwd(f):
    R, color, others, radii, pixels, gb, bb, ib = f
    return _C.rasterize_gaussians_backward(t["bg"], t["means3D"], radii, t["colors"], t["scales"], t["rotations"], 1.0, e, t["viewmatrix"], eye,
                                           t["beams"], gc, go, e, 1, z3, gb, R, bb, ib, False)


for _ in range(3):
    f = fwd(); g = bwd(f)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    f = fwd()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(iters):
    g = bwd(f)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"surfel P={P} {H}x{W}: instances={f[0]} visible={(f[3] > 0).sum().item()} fwd {1e3 * (t1 - t0) / iters:.3f} ms  bwd {1e3 * (t2 - t1) / iters:.3f} ms  "
      f"-> {iters / ((t2 - t0)):.1f} fwd+bwd frames/s")
print("alpha mean", f[2][1].mean().item(), "finite grads", all(torch.isfinite(x).all().item() for x in g))
