"""Randomized check of the enqueue-only forward (lidargs_forward_enqueue: no host read, binning capacity learnt from the previous
frame) and of its HIP-graph replay against the oracle: for every random scene the module renders once the default way, then
enqueue-only (forward + backward compared with the oracle), then the same as a captured graph replayed twice.

    python tools/enqueue_sweep.py [first_seed] [n] > profiles/rNN_enqueue_sweep.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import lidargs_scenes as sc
import util
from util import GRAD_KEYS_SR, make_settings, oracle_forward_backward, parity, to_torch
from diff_lidargs_rasterization import GaussianRasterizer

first = int(sys.argv[1]) if len(sys.argv) > 1 else 11000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
failed, scenes = [], 0
t0 = time.time()
NAMES = ("means3D", "colors", "opacities", "scales", "rotations")
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    H = int(rng.choice([2, 3, 16, 17, 32, 64])); W = int(rng.integers(1, 900)); P = int(rng.integers(1, 20000))
    kind = "shell" if rng.random() < 0.5 else "street"
    beams = str(rng.choice(["uniform", "waymo", "neartie"])) if H >= 4 else "uniform"
    desc = dict(seed=seed, kind=kind, P=P, H=H, W=W, beams=beams)
    scenes += 1
    try:
        scene = sc.make_scene(kind, P, H, seed % 1000, random_view=True, beams=beams)
        grads = sc.upstream_grads(H, W, seed % 1000)
        ref = oracle_forward_backward(scene, W, H, grads)
        st = to_torch(scene)
        gc, gd, go = (torch.from_numpy(g).cuda() for g in grads)
        rast = GaussianRasterizer(make_settings(st, W, H))
        rast.enqueue_only = True
        leaves = [st[k].clone().requires_grad_(True) for k in NAMES]
        m2 = torch.zeros((P, 4), device="cuda", requires_grad=True)

        def frame():
            for t in leaves + [m2]:
                t.grad = None
            c, d, o, r = rast(means3D=leaves[0], means2D=m2, opacities=leaves[2], colors_precomp=leaves[1], scales=leaves[3], rotations=leaves[4])
            torch.autograd.backward([c, d, o], [gc, gd, go])
            return c, d, o, r

        def check(tag, out):
            c, d, o, r = out
            assert np.array_equal(r.cpu().numpy(), ref["radii"]), tag + ": radii"
            parity(tag + ".color", c.detach().cpu().numpy(), ref["color"], verbose=False)
            parity(tag + ".depth", d.detach().cpu().numpy(), ref["depth"], verbose=False)
            parity(tag + ".occ", o.detach().cpu().numpy(), ref["occ"], verbose=False)
            got = dict(dL_dmeans3D=leaves[0].grad, dL_dmeans2D=m2.grad, dL_dcolors=leaves[1].grad, dL_dopacity=leaves[2].grad,
                       dL_dscales=leaves[3].grad, dL_drotations=leaves[4].grad)
            for k in GRAD_KEYS_SR:
                parity(tag + "." + k, got[k].cpu().numpy(), ref[k], verbose=False)
        frame()                                   # the default forward: learns the capacity
        check("enqueue", frame())                 # enqueue-only
        assert not rast.enqueue_status()["overflow"], "overflow on an unchanged scene"
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            frame()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        for t in leaves + [m2]:
            t.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = frame()
        g.replay(); g.replay(); torch.cuda.synchronize()
        check("graph", out)
        del g
    except AssertionError as e:
        failed.append(dict(failed=str(e)[:300], **desc))
    except Exception as e:
        failed.append(dict(failed="EXCEPTION " + repr(e)[:300], **desc))
log = util.PARITY_LOG
img = [f for f in failed if any(x in f["failed"] for x in (".color", ".depth", ".occ", "radii", "EXCEPTION", "overflow"))]
print(json.dumps({"what": "tools/enqueue_sweep.py: enqueue-only forward + backward and its HIP-graph replay vs the oracle on random scenes",
                  "scenes": scenes, "seconds": round(time.time() - t0, 1), "first_seed": first, "parity_calls": len(log),
                  "entries_compared": int(sum(s["n"] for s in log)), "failed_on_image_radii_or_exception": img,
                  "failed_on_a_gradient_budget": [f for f in failed if f not in img]}, indent=1))
