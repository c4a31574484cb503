"""world_size-2 (and 3) gloo tests of the range-shell multi-GPU path on CPU.

The product code under test is lidargs_dist._ShellRasterize (shell edges, the two all-gathers, the
transmittance products, T_final selection, behind-sums, gradient reduce-scatter).  The per-rank
renderer is the oracle-backed stand-in of tests/dist_backend_oracle.py, and the expected result is
the plain single-process oracle on the whole scene."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import lidargs_scenes as sc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _settings(scene, W, H):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")]
    from diff_lidargs_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    return GaussianRasterizationSettings(H, W, 1.0, 1.0, t(scene["bg"]), 1.0, t(scene["viewmatrix"]), torch.eye(4), 1, torch.zeros(3), False,
                                         t(scene["beams"]), 80, 0, False)


def _worker(rank, world, port, kind, P, H, W, seed, bg, grad_sync, outdir, edges=None, wedges=False):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lidargs_dist
    from dist_backend_oracle import OracleShellBackend, OracleWedgeBackend
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    scene["bg"] = np.array(bg, np.float32)
    if wedges:
        rast = lidargs_dist.WedgeRasterizer(_settings(scene, W, H), lidargs_dist.TorchDistComm(), OracleWedgeBackend(), grad_sync=grad_sync, edges=edges)
    else:
        rast = lidargs_dist.ShellRasterizer(_settings(scene, W, H), lidargs_dist.TorchDistComm(), OracleShellBackend(), grad_sync=grad_sync,
                                            edges=None if edges is None else torch.tensor(edges, dtype=torch.float32))
    rast.ship_live = seed % 2 == 1          # round 6: both forms of the gradient exchange (live rows only / every selected row); "auto" would pick the second at these sizes
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(True)
    leaves = {k: t(scene[k]) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    m2 = torch.zeros(P, 4, requires_grad=True)
    gc, gd, go = (torch.from_numpy(g) for g in sc.upstream_grads(H, W, seed))
    if grad_sync == "shard":
        # the rank's own rows as leaves; the replicated tensors only feed the rendering.  Saved as dense [P, w] arrays (zeros outside the
        # chunk) so that the checks of "reduce_scatter" apply unchanged.
        base, n = lidargs_dist.shard_rows(P, world, rank)
        shard = {k: leaves[k].detach()[base:base + n].clone().requires_grad_(True) for k in leaves}
        shard["means2D"] = torch.zeros(n, 4, requires_grad=True)
        color, depth, occ, radii = rast(leaves["means3D"], m2, leaves["opacities"], leaves["colors"], leaves["scales"], leaves["rotations"], shard=shard)
        torch.autograd.backward([color, depth, occ], [gc, gd, go])
        assert all(leaves[k].grad is None for k in leaves) and m2.grad is None
        for k, full in list(leaves.items()) + [("means2D", m2)]:
            assert tuple(shard[k].grad.shape) == (n,) + tuple(full.shape[1:])
            full.grad = torch.zeros_like(full)
            full.grad[base:base + n] = shard[k].grad
    else:
        color, depth, occ, radii = rast(leaves["means3D"], m2, leaves["opacities"], leaves["colors"], leaves["scales"], leaves["rotations"])
        torch.autograd.backward([color, depth, occ], [gc, gd, go])
    out = dict(color=color.detach().numpy(), depth=depth.detach().numpy(), occ=occ.detach().numpy(), radii=radii.numpy(),
               dL_dmeans3D=leaves["means3D"].grad.numpy(), dL_dmeans2D=m2.grad.numpy(), dL_dcolors=leaves["colors"].grad.numpy(),
               dL_dopacity=leaves["opacities"].grad.numpy(), dL_dscales=leaves["scales"].grad.numpy(),
               dL_drotations=leaves["rotations"].grad.numpy())
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    ("w2_shell", 2, "shell", 4000, 16, 256, 31, (0.0, 0.0), "all_reduce"),
    ("w2_street_bg", 2, "street", 6000, 16, 256, 32, (0.3, 0.6), "reduce_scatter"),
    ("w3_dense", 3, "street", 9000, 16, 128, 33, (0.1, 0.2), "reduce_scatter"),   # saturating pixels: the T<1e-4 stop crosses shells
    ("w2_dense_rs", 2, "shell", 3001, 16, 256, 34, (0.0, 0.1), "reduce_scatter_dense"),   # P not divisible by the world size
    ("w3_odd_sparse", 3, "shell", 2999, 16, 256, 35, (0.2, 0.0), "reduce_scatter"),
    ("w2_none", 2, "shell", 3000, 16, 256, 36, (0.0, 0.0), "none"),
    ("w3_shard", 3, "street", 5000, 16, 256, 37, (0.1, 0.3), "shard"),             # round 6: the rank's own chunk of every gradient, [rows_r, w]
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_shell_sharding_matches_single_process(case, tmp_path):
    from util import GRAD_KEYS_SR, oracle_forward_backward, parity
    name, world, kind, P, H, W, seed, bg, grad_sync = case
    mp.spawn(_worker, args=(world, _free_port(), kind, P, H, W, seed, bg, grad_sync, str(tmp_path)), nprocs=world, join=True)
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    scene["bg"] = np.array(bg, np.float32)
    ref = oracle_forward_backward(scene, W, H, sc.upstream_grads(H, W, seed))
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    # the image is identical on every rank and matches the single-process composite
    for r in range(world):
        np.testing.assert_array_equal(ranks[r]["radii"], ref["radii"])
        for k in ("color", "depth", "occ"):
            parity(f"{k}@rank{r}", ranks[r][k], ref[k], verbose=(r == 0))
            np.testing.assert_array_equal(ranks[r][k], ranks[0][k])
    # gradients: all_reduce -> full on every rank; reduce_scatter -> rank r holds rows [r*rows, (r+1)*rows)
    rows = (P + world - 1) // world
    for k in GRAD_KEYS_SR:
        if grad_sync == "all_reduce":
            full = ranks[0][k]
            np.testing.assert_array_equal(ranks[1][k], full)
        elif grad_sync == "none":
            # every rank keeps its own shell's rows: supports are disjoint and their union is the full gradient
            nz = [np.abs(ranks[r][k]).reshape(P, -1).max(1) > 0 for r in range(world)]
            assert not (nz[0] & nz[1]).any()
            full = sum(ranks[r][k] for r in range(world))
        else:
            full = np.zeros_like(ref[k])
            for r in range(world):
                sl = slice(r * rows, min(P, (r + 1) * rows))
                full[sl] = ranks[r][k][sl]
                outside = np.ones(P, bool); outside[sl] = False
                assert float(np.abs(ranks[r][k][outside]).max(initial=0.0)) == 0.0
        parity(k, full, ref[k])


def test_empty_shells_are_harmless(tmp_path):
    """Explicit edges that leave the first and the last of three shells empty."""
    from util import GRAD_KEYS_SR, oracle_forward_backward, parity
    world, kind, P, H, W, seed, bg = 3, "shell", 2000, 16, 256, 37, (0.1, 0.0)
    edges = [float("-inf"), 1e-3, 1e6, float("inf")]
    mp.spawn(_worker, args=(world, _free_port(), kind, P, H, W, seed, bg, "reduce_scatter", str(tmp_path), edges), nprocs=world, join=True)
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    scene["bg"] = np.array(bg, np.float32)
    ref = oracle_forward_backward(scene, W, H, sc.upstream_grads(H, W, seed))
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    rows = (P + world - 1) // world
    for r in range(world):
        for k in ("color", "depth", "occ"):
            parity(f"{k}@rank{r}", ranks[r][k], ref[k], verbose=False)
    for k in GRAD_KEYS_SR:
        full = np.zeros_like(ref[k])
        for r in range(world):
            sl = slice(r * rows, min(P, (r + 1) * rows))
            full[sl] = ranks[r][k][sl]
        parity(k, full, ref[k], verbose=False)


def test_shell_edges_balance_and_cover():
    import lidargs_dist
    scene = sc.make_scene("street", 50_000, 16, 5)
    e = lidargs_dist.shell_edges(torch.from_numpy(scene["means3D"]), torch.from_numpy(scene["viewmatrix"]), 4, 0, 80)
    assert e.shape == (5,) and e[0] == float("-inf") and e[-1] == float("inf")
    assert torch.all(e[1:] > e[:-1])
    r = np.linalg.norm(scene["means3D"], axis=1)
    inside = r < 80
    counts = np.histogram(r[inside], bins=np.array([-1.0] + e[1:-1].tolist() + [1e9]))[0]
    assert counts.min() > 0.8 * inside.sum() / 4 and counts.max() < 1.2 * inside.sum() / 4


def test_single_comm_world_of_one_equals_plain_oracle():
    """The shell machinery with one rank degenerates to the plain single-GPU walk."""
    import lidargs_dist
    from dist_backend_oracle import OracleShellBackend
    from util import oracle_forward_backward, parity
    kind, P, H, W, seed = "shell", 3000, 16, 256, 41
    scene = sc.make_scene(kind, P, H, seed)
    scene["bg"] = np.array([0.2, 0.1], np.float32)
    rast = lidargs_dist.ShellRasterizer(_settings(scene, W, H), lidargs_dist.SingleComm(), OracleShellBackend())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(True)
    lv = {k: t(scene[k]) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    m2 = torch.zeros(P, 4, requires_grad=True)
    color, depth, occ, radii = rast(lv["means3D"], m2, lv["opacities"], lv["colors"], lv["scales"], lv["rotations"])
    grads = sc.upstream_grads(H, W, seed)
    torch.autograd.backward([color, depth, occ], [torch.from_numpy(g) for g in grads])
    ref = oracle_forward_backward(scene, W, H, grads)
    parity("color", color.detach().numpy(), ref["color"]); parity("depth", depth.detach().numpy(), ref["depth"])
    parity("dL_dmeans3D", lv["means3D"].grad.numpy(), ref["dL_dmeans3D"])
    parity("dL_dopacity", lv["opacities"].grad.numpy(), ref["dL_dopacity"])


# ---- column wedges -------------------------------------------------------------------------------------------------------------
WEDGE_CASES = [
    ("w2_shell", 2, "shell", 4000, 16, 256, 61, (0.0, 0.0), "all_reduce", None),
    ("w2_street_bg", 2, "street", 6000, 16, 256, 62, (0.3, 0.6), "reduce_scatter", None),
    ("w3_dense", 3, "street", 9000, 16, 160, 63, (0.1, 0.2), "reduce_scatter", None),
    ("w3_ragged_odd", 3, "shell", 2999, 18, 250, 64, (0.2, 0.0), "reduce_scatter", None),        # W % 16 != 0, P % world != 0
    ("w4_narrow_first_wedge", 4, "shell", 3000, 16, 256, 65, (0.0, 0.1), "reduce_scatter", [0, 16, 128, 240, 256]),   # one-tile wedges
    ("w3_shard", 3, "street", 5001, 16, 250, 66, (0.2, 0.1), "shard", None),                  # round 6 (boundary Gaussians' rows are ADDED into the chunk)
    ("w2_none", 2, "shell", 3000, 16, 256, 66, (0.0, 0.0), "none", None),
]


@pytest.mark.parametrize("case", WEDGE_CASES, ids=[c[0] for c in WEDGE_CASES])
def test_wedge_sharding_matches_single_process(case, tmp_path):
    """lidargs_dist._WedgeRasterize over gloo: edges, selection bound, image gather, radii max-reduce, gradient all-to-all with
    addition of the boundary Gaussians' partial rows -- against the plain single-process oracle.  The image must be EXACTLY the
    single-process one (a rank's lists are the complete lists of its tiles)."""
    from util import GRAD_KEYS_SR, oracle_forward_backward, parity
    name, world, kind, P, H, W, seed, bg, grad_sync, edges = case
    mp.spawn(_worker, args=(world, _free_port(), kind, P, H, W, seed, bg, grad_sync, str(tmp_path), edges, True), nprocs=world, join=True)
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    scene["bg"] = np.array(bg, np.float32)
    ref = oracle_forward_backward(scene, W, H, sc.upstream_grads(H, W, seed))
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    for r in range(world):
        np.testing.assert_array_equal(ranks[r]["radii"], ref["radii"])
        for k in ("color", "depth", "occ"):
            np.testing.assert_array_equal(ranks[r][k], ref[k])          # bit for bit
    rows = (P + world - 1) // world
    for k in GRAD_KEYS_SR:
        if grad_sync == "all_reduce":
            full = ranks[0][k]
            np.testing.assert_array_equal(ranks[1][k], full)
        elif grad_sync == "none":
            full = sum(ranks[r][k] for r in range(world))                # partial sums of the wedges' pixels
        else:
            full = np.zeros_like(ref[k])
            for r in range(world):
                sl = slice(r * rows, min(P, (r + 1) * rows))
                full[sl] = ranks[r][k][sl]
                outside = np.ones(P, bool); outside[sl] = False
                assert float(np.abs(ranks[r][k][outside]).max(initial=0.0)) == 0.0
        parity(k, full, ref[k])     # dL_dmeans2D[:, 2] is a sum of per-pixel norms (additive over wedges), the rest linear in the pixel sums


def test_wedge_edges_cover_and_balance():
    import lidargs_dist
    scene = sc.make_scene("street", 50_000, 16, 5)
    for world, W in ((4, 2650), (8, 2650), (3, 250), (16, 256)):
        e = lidargs_dist.wedge_edges(torch.from_numpy(scene["means3D"]), torch.from_numpy(scene["viewmatrix"]), W, world,
                                     scales=torch.from_numpy(scene["scales"]))
        assert len(e) == world + 1 and e[0] == 0 and e[-1] == W
        assert all(b > a for a, b in zip(e, e[1:])) and all(x % 16 == 0 for x in e[1:-1])
    with pytest.raises(ValueError):
        lidargs_dist.wedge_edges(torch.from_numpy(scene["means3D"]), torch.from_numpy(scene["viewmatrix"]), 64, 8)
