"""Stage times of small frames (the per-rank / small-scene launch floor):  python tools/small_frames.py"""
import sys, time, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/lidar-gs_amd", "/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch, make_settings
from diff_lidargs_rasterization import GaussianRasterizer, _C
for kind, P, H, W, seed in [("shell", 10000, 16, 512, 1), ("street", 60000, 64, 2650, 3), ("street", 250000, 64, 2650, 3)]:
    scene = sc.make_scene(kind, P, H, seed); st = to_torch(scene)
    rast = GaussianRasterizer(make_settings(st, W, H))
    m2 = torch.zeros(P, 4, device="cuda", requires_grad=True)
    leaves = [st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")]
    gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed))
    def step():
        c, d, o, r = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], colors_precomp=leaves[1], scales=leaves[3], rotations=leaves[4])
        torch.autograd.backward([c, d, o], [gc, gd, go])
    for _ in range(20): step()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(200): step()
    torch.cuda.synchronize(); dt = (time.time() - t) / 200
    _C.profile_enable(True)
    for _ in range(20): step()
    torch.cuda.synchronize(); _C.profile_enable(False)
    s = _C.profile_summary()
    print(f"{kind} P={P} {H}x{W}: ms/frame {dt * 1e3:.3f}; stage sum {sum(v[0] for v in s.values()):.3f}", {k: round(v[0], 3) for k, v in s.items()}, {k: _C.last_counters()[k] for k in ("instances", "tile_rows", "segments")})
