"""Time the fused anchor decode (forward + backward) against the same computation as a chain of framework ops on the same GPU.
    python tools/time_decode.py [N anchors] [k] [iters]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import lidargs_scenes as _sc
build_pc, random_case = _sc.anchor_model_to_torch, _sc.make_anchor_model
from neural_gaussians import generate_neural_gaussians
from oracle import neural_gaussians_torch as ngt

N = int(sys.argv[1]) if len(sys.argv) > 1 else 333_334
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
p, cam, vis, rng = random_case(N, k, 5)
pc = build_pc(p)
camera = types.SimpleNamespace(camera_center=torch.from_numpy(cam).cuda(), uid=0)
vmask = torch.from_numpy(vis).cuda()
params = {m: tuple(getattr(pc, "mlp_" + m)[i].weight if j == 0 else getattr(pc, "mlp_" + m)[i].bias for i in (0, 2) for j in (0, 1)) for m in ("opacity", "cov", "color", "raydrop")}
flags = (p["add_opacity_dist"], p["add_cov_dist"], p["add_color_dist"])
leaves = [pc._anchor_feat, pc._anchor, pc._offset, pc.get_scaling] + [t for m in params.values() for t in m]


def run(fn):
    for t in leaves: t.grad = None
    outs = fn()
    xyz, color, opacity, scaling, rot = outs[:5]
    loss = xyz.sum() + color.sum() + opacity.sum() + scaling.sum() + rot.sum()
    loss.backward()
    return xyz.shape[0]


def timeit(fn, fwd_only=False):
    for _ in range(3):
        (fn() if fwd_only else run(fn))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        (fn() if fwd_only else run(fn))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters


hip = lambda: generate_neural_gaussians(camera, pc, vmask, is_training=True)
eager = lambda: ngt.generate(pc._anchor_feat, pc._anchor, pc._offset, pc.get_scaling, params, camera.camera_center, vmask, flags)
M = run(hip)
if len(sys.argv) > 4 and sys.argv[4] == "hip":          # the fused path alone (profiling runs: tools/pmc_decode.sh)
    print(f"anchor decode N={N} k={k}: {M} Gaussians out; forward+backward HIP {timeit(hip):.3f} ms")
    sys.exit(0)
with torch.no_grad():
    f_hip, f_eager = timeit(hip, True), timeit(eager, True)
b_hip, b_eager = timeit(hip), timeit(eager)
print(f"anchor decode N={N} k={k}: {M} Gaussians out; forward HIP {f_hip:.3f} ms vs framework ops {f_eager:.3f} ms; "
      f"forward+backward HIP {b_hip:.3f} ms vs framework ops {b_eager:.3f} ms")
