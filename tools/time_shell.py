"""Per-rank cost of the range-shell path at world N, measured on ONE GPU: the N virtual ranks first run as threads with an
in-memory communicator that records every collective's result; then each rank is replayed alone (collectives return the
recorded tensors instantly) and timed.  Gives compute + host glue per rank, i.e. the frame time at world N minus RCCL time.

    python tools/time_shell.py [world] [cfg] [iters]          (MODE=wedge in the environment: column wedges instead of range shells)
"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import lidargs_dist
import lidargs_scenes as sc
from test_dist_gpu import ThreadComm
from util import make_settings, to_torch

MODE = os.environ.get("MODE", "shell")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kind, P, H, W, seed = sc.BASELINE_CONFIGS[cfg]
scene = sc.make_scene(kind, P, H, seed)
st = to_torch(scene)
gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed))
settings = make_settings(st, W, H)
import math
tile_rad = (16 * 2 * math.pi / W, 4 * float(st["beams"][-1] - st["beams"][0]) / max(1, H - 1))



class RecordComm(ThreadComm):
    def __init__(self, shared, rank, log):
        super().__init__(shared, rank); self.log = log

    def _rec(self, name, out):
        self.log.append((name, out.clone())); return out

    def all_gather(self, t): return self._rec("all_gather", super().all_gather(t))
    def all_reduce(self, t): return self._rec("all_reduce", super().all_reduce(t))
    def all_reduce_async(self, t):
        self.all_reduce(t); return lambda: None
    def all_reduce_max_async(self, t):
        super().all_reduce_max_async(t); self.log.append(("all_reduce", t.clone())); return lambda: None
    def all_to_all_rows(self, t, s, r): return self._rec("all_to_all_rows", super().all_to_all_rows(t, s, r))
    def reduce_scatter_rows(self, t): return self._rec("reduce_scatter_rows", super().reduce_scatter_rows(t))


class ReplayComm:
    def __init__(self, rank, world, log):
        self.rank, self.world, self.log, self.i = rank, world, log, 0

    def _next(self, name):
        n, out = self.log[self.i % len(self.log)]; self.i += 1
        assert n == name, (n, name)
        return out

    def all_gather(self, t): return self._next("all_gather")
    def all_reduce(self, t): t.copy_(self._next("all_reduce")); return t
    def all_reduce_async(self, t):
        self.all_reduce(t); return lambda: None
    def all_reduce_max_async(self, t):
        self.all_reduce(t); return lambda: None
    def all_to_all_rows(self, t, s, r): return self._next("all_to_all_rows")
    def reduce_scatter_rows(self, t): return self._next("reduce_scatter_rows")
    def broadcast(self, t, src=0): return t


def frame(mod):
    if MODE == "wedge":
        outs, saved = lidargs_dist.wedge_forward(mod, st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"])
        return lidargs_dist.wedge_backward(mod, saved, gc, gd, go)
    outs, saved = lidargs_dist.shell_forward(mod, st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"])
    return lidargs_dist.shell_backward(mod, saved, gc, gd, go)


def make_module(comm, edges):
    mod = (lidargs_dist.WedgeRasterizer if MODE == "wedge" else lidargs_dist.ShellRasterizer)(settings, comm, edges=edges,
                                                                                              grad_sync=os.environ.get("GRAD_SYNC", "reduce_scatter"))   # GRAD_SYNC=shard: the rank's own chunk only
    mod.enqueue_only = os.environ.get("ENQUEUE", "0") == "1"      # enqueue-only rank frames (no host read after the first frame)
    return mod


def measure(edges):
    """(per-rank wall ms without RCCL, per-rank sum of kernel-stage ms) for the given cut, by record + replay."""
    from diff_lidargs_rasterization import _C
    logs = [[] for _ in range(world)]
    shared = ThreadComm.Shared(world)
    errs = []

    def record(r):
        comm = None
        try:
            torch.cuda.set_device(0)
            comm = RecordComm(shared, r, logs[r])
            frame(make_module(comm, edges))
        except Exception as e:
            errs.append(e); shared.barrier.abort()
        finally:
            if comm is not None:
                comm.finish()

    th = [threading.Thread(target=record, args=(r,)) for r in range(world)]
    for t in th: t.start()
    for t in th: t.join(timeout=300)
    assert not errs, errs
    torch.cuda.synchronize()
    walls, stages = [], []
    only = os.environ.get("ONLY_RANK")                     # replay one rank only (for a kernel trace of its frame)
    for r in (range(world) if only is None else [int(only)]):
        mod = make_module(ReplayComm(r, world, logs[r]), edges)
        for _ in range(3):
            frame(mod)
        torch.cuda.synchronize()
        prof = None
        if os.environ.get("CPROFILE"):                     # where the HOST's time goes inside the timed loop
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        t0 = time.perf_counter()
        for _ in range(iters):
            frame(mod)
        torch.cuda.synchronize()
        walls.append(1e3 * (time.perf_counter() - t0) / iters)
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats("tottime").print_stats(28)
        if os.environ.get("NOSTAGE"):
            stages.append(0.0); del mod
            continue
        _C.profile_enable(True)
        for _ in range(iters):
            frame(mod)
        torch.cuda.synchronize()
        _C.profile_enable(False)
        stages.append(sum(v[0] for v in _C.profile_summary().values()))
        del mod
    return walls, stages, logs


shares = [1.0 / world] * world
rounds = int(os.environ.get("TUNE", "0"))
for it in range(rounds + 1):
    if MODE == "wedge":
        edges = lidargs_dist.wedge_edges(st["means3D"], st["viewmatrix"], W, world, scales=st["scales"], shares=shares)
    else:
        edges = lidargs_dist.shell_edges(st["means3D"], st["viewmatrix"], world, 0, 80, scales=st["scales"] if os.environ.get("EDGES", "w") == "w" else None,
                                         tile_rad=tile_rad, shares=shares)
    walls, stages, logs = measure(edges)
    print(f"{MODE} world {world} {cfg} round {it}: edges " + " ".join(f"{float(e):.1f}" for e in edges[1:-1]))
    print("   per-rank ms (no RCCL time): " + " ".join(f"{t:.3f}" for t in walls) + f"  max {max(walls):.3f}")
    print("   per-rank kernel-stage ms:   " + " ".join(f"{t:.3f}" for t in stages) + f"  max {max(stages):.3f}")
    shares = lidargs_dist.rebalance_shares(shares, stages, fixed=float(os.environ.get("FIXED", "0.25")))
coll = {}
for n, t in logs[0]:
    coll.setdefault(n, []).append(t.numel() * t.element_size())
print("collectives per frame (bytes of the result on a rank):", {k: v for k, v in coll.items()})
import json
out = os.environ.get("OUT_JSON")
if out:
    json.dump({"what": "per-rank cost of the range-shell path, each virtual rank replayed alone on one MI355X (collectives return recorded "
                       "tensors instantly: compute + host glue per rank, no RCCL time)", "sharding": MODE, "world": world, "workload": cfg, "iters": iters,
               "edges": [float(e) for e in edges[1:-1]], "per_rank_wall_ms_without_rccl": walls, "per_rank_kernel_stage_ms": stages,
               "collective_result_bytes_per_frame": coll}, open(out, "w"), indent=1)
