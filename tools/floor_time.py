"""Diagnostic: stage times of cfg3 with almost everything culled (lidar_far small): the launch / empty-block floor of each stage."""
import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/lidar-gs_amd", "/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch
from diff_lidargs_rasterization import GaussianRasterizer, _C
kind, P, H, W, seed = sc.BASELINE_CONFIGS["cfg3"]
far = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scene = sc.make_scene(kind, P, H, seed); st = to_torch(scene)
gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed))
rast = GaussianRasterizer(sc.raster_settings(st, W, H, far, 0, 1.0, False))
leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
means2D = torch.zeros((P, 4), device="cuda", requires_grad=True)
def step():
    for t in list(leaves.values()) + [means2D]: t.grad = None
    c, d, o, r = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
    torch.autograd.backward([c, d, o], [gc, gd, go])
for _ in range(5): step()
_C.profile_enable(True)
for _ in range(20): step()
torch.cuda.synchronize()
print(_C.last_counters())
print({k: round(v[0], 4) for k, v in _C.profile_summary().items()})
