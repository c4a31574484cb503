"""A frame far above the BASELINE sizes (maximum-size behaviour):  python tools/big_frame.py [P_millions] [H] [W]
Forward + backward through the drop-in package; prints time, instance count, finiteness, and the size-independent properties the
full-size tests use (occ = 1 - T within [0, 1], zero gradient rows for culled Gaussians, determinism of the forward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import lidargs_scenes as sc
from diff_lidargs_rasterization import GaussianRasterizer, _C
P = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 32_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
kind = sys.argv[4] if len(sys.argv) > 4 else "shell"
t0 = time.time(); scene = sc.make_scene(kind, P, H, 9); print("scene", round(time.time() - t0, 1), "s", flush=True)
st = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in scene.items()}
rast = GaussianRasterizer(sc.raster_settings(st, W, H))
leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
m2 = torch.zeros(P, 4, device="cuda", requires_grad=True)
g = [torch.from_numpy(x).cuda() for x in sc.upstream_grads(H, W, 9)]
def frame():
    for t in list(leaves.values()) + [m2]: t.grad = None
    c, d, o, r = rast(means3D=leaves["means3D"], means2D=m2, opacities=leaves["opacities"], colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
    torch.autograd.backward([c, d, o], g)
    return c, d, o, r
c, d, o, r = frame(); torch.cuda.synchronize()
c2, d2, o2, r2 = frame(); torch.cuda.synchronize()
t0 = time.time()
for _ in range(5): frame()
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
cnt = _C.last_counters()
vis = r > 0
print(dict(P=P, H=H, W=W, ms=round(dt * 1e3, 3), instances=cnt["instances"], tile_rows=cnt["tile_rows"], segments=cnt["segments"], visible=int(vis.sum()),
           deterministic=bool(torch.equal(c, c2) and torch.equal(d, d2) and torch.equal(r, r2)), occ_range=(float(o.min()), float(o.max())),
           finite=all(bool(torch.isfinite(t.grad).all()) for t in list(leaves.values()) + [m2]),
           culled_rows_zero=bool((leaves["means3D"].grad[~vis] == 0).all() and (leaves["scales"].grad[~vis] == 0).all()),
           touched_rows=int((leaves["opacities"].grad.view(-1) != 0).sum()), mem_GB=round(torch.cuda.max_memory_allocated() / 2**30, 2)))
