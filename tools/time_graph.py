"""Frame time of the enqueue-only forward + backward replayed from a HIP graph, next to the eager frame (ordinary and enqueue-only).

    python tools/time_graph.py [cfg] [iters]
"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import lidargs_scenes as sc
from diff_lidargs_rasterization import GaussianRasterizer
from util import make_settings, to_torch

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kind, P, H, W, seed = sc.BASELINE_CONFIGS[cfg]
st = to_torch(sc.make_scene(kind, P, H, seed))
grads = [torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed)]
leaves = [st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")]
means2D = torch.zeros((P, 4), device="cuda", requires_grad=True)
inputs = leaves + [means2D]


def frame(rast):
    for t in inputs:
        t.grad = None
    out = rast(means3D=leaves[0], means2D=means2D, opacities=leaves[2], colors_precomp=leaves[1], scales=leaves[3], rotations=leaves[4])
    torch.autograd.backward(list(out[:3]), grads)
    return out


def timed(fn, n):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
res = {}
with torch.cuda.stream(side):
    plain = GaussianRasterizer(make_settings(st, W, H))
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        frame(plain)
    res["eager lidargs_forward"] = timed(lambda: frame(plain), iters)
    rast = GaussianRasterizer(make_settings(st, W, H))
    rast.enqueue_only = True
    res["eager enqueue-only"] = timed(lambda: frame(rast), iters)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for t in inputs:
    t.grad = None
gc.collect()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    frame(rast)
res["graph replay"] = timed(graph.replay, iters)
assert not rast.enqueue_status()["overflow"]
for k, v in res.items():
    print(f"{cfg} {k:28s} {v:.4f} ms/frame")
