"""Randomized sweep of the SHARDED paths (range shells, column wedges) on virtual ranks -- N threads on one GPU with an in-memory
communicator, tests/test_dist_gpu.py's harness -- against the plain single-GPU HIP path and the CPU oracle.

    python tools/dist_sweep.py [first_seed] [n] > profiles/rNN_dist_sweep.json

Random world sizes (2..8), cut, gradient exchange, scene kind, image size (narrow ones included), beam table and background."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import lidargs_scenes as sc
import util
from util import GRAD_KEYS_SR, hip_forward_backward, oracle_forward_backward, parity
from test_dist_gpu import _assemble, _virtual_ranks

first = int(sys.argv[1]) if len(sys.argv) > 1 else 7000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
scenes, failed = [], []
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    H = int(rng.choice([2, 3, 16, 17, 32, 40, 64])); W = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 900))]))
    P = int(rng.integers(50, 20000))
    world = int(rng.choice([2, 3, 4, 5, 8]))
    wedges = bool(rng.integers(0, 2)) and (W + 15) // 16 >= world      # (fewer tile columns than ranks: the wedge path refuses, loudly)
    kind = "shell" if rng.random() < 0.5 else "street"
    beams = str(rng.choice(["uniform", "waymo", "neartie"])) if H >= 4 else "uniform"
    sync = str(rng.choice(["all_reduce", "reduce_scatter", "shard"]))      # round 6: + the rank's own chunk only; _run_rank runs the live-row exchange on odd P
    enqueue = bool(rng.random() < 0.34)                                # enqueue-only rank frames: an ordinary frame, then two that read nothing back
    desc = dict(seed=seed, cut="wedges" if wedges else "shells", world=world, kind=kind, P=P, H=H, W=W, beams=beams, grad_sync=sync,
                enqueue_only=enqueue)
    n0 = len(util.PARITY_LOG)
    try:
        scene = sc.make_scene(kind, P, H, seed % 1000, random_view=True, beams=beams)
        scene["bg"] = np.array([rng.random() * 0.5, rng.random() * 0.5], np.float32) if rng.random() < 0.5 else scene["bg"]
        grads = sc.upstream_grads(H, W, seed % 1000)
        plain = hip_forward_backward(scene, W, H, grads)
        ref = oracle_forward_backward(scene, W, H, grads)
        results = _virtual_ranks(world, scene, W, H, grads, sync, wedges, frames=3 if enqueue else 1, enqueue=enqueue)
        ident = True
        for r in range(world):
            assert np.array_equal(results[r]["radii"], plain["radii"]), f"radii differ on rank {r}"
            for k in ("color", "depth", "occ"):
                ident = ident and np.array_equal(results[r][k], plain[k])
                parity(f"{k}@rank{r} vs plain", results[r][k], plain[k], verbose=False)
                parity(f"{k}@rank{r}", results[r][k], ref[k], verbose=False)
        desc["image_bit_identical_to_single_gpu"] = bool(ident)
        full = _assemble(results, ref, P, world, sync)
        for k in GRAD_KEYS_SR:
            parity(f"{k} vs plain", full[k], plain[k], verbose=False)
            parity(k, full[k], ref[k], verbose=False)
    except AssertionError as e:
        desc["failed"] = str(e)[:300]; failed.append(desc)
    except Exception as e:                                             # a crash of a virtual rank is a finding too
        desc["failed"] = "EXCEPTION " + repr(e)[:300]; failed.append(desc)
    log = util.PARITY_LOG[n0:]
    desc["entries"] = int(sum(s["n"] for s in log))
    scenes.append(desc)
log = util.PARITY_LOG
print(json.dumps({
    "what": "tools/dist_sweep.py: sharded frames on virtual ranks (threads on one GPU, in-memory collectives) vs the single-GPU HIP path and the oracle",
    "scenes": len(scenes), "seconds": round(time.time() - t0, 1), "first_seed": first,
    "by_cut": {c: sum(1 for s in scenes if s["cut"] == c) for c in ("shells", "wedges")},
    "scenes_with_enqueue_only_rank_frames": sum(1 for s in scenes if s.get("enqueue_only")),
    "by_world": {str(w): sum(1 for s in scenes if s["world"] == w) for w in (2, 3, 4, 5, 8)},
    "narrow_images_W_below_40": sum(1 for s in scenes if s["W"] < 40),
    "wedge_scenes_bit_identical_to_single_gpu": sum(1 for s in scenes if s["cut"] == "wedges" and s.get("image_bit_identical_to_single_gpu")),
    "parity_calls": len(log), "entries_compared": int(sum(s["n"] for s in log)),
    "soft_entries": int(sum(s.get("soft", 0) for s in log)), "flip_entries": int(sum(s.get("flips", 0) for s in log)),
    "failed_scenes": failed}, indent=1))
