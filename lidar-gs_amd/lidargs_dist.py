"""lidargs_dist -- one scene, N GPUs of one node: Gaussians sharded by RANGE SHELL.

The reference is single-GPU; this is new design, shaped by the one property of the path that
matters: front-to-back compositing is order-dependent and every tile's order is by range, so if
rank g owns the Gaussians with range in [e_g, e_{g+1}) each pixel's sorted list is the
concatenation of the ranks' lists.  Per-Gaussian work and gradients are then LOCAL to one rank;
only per-pixel planes (and, if the caller wants index-sharded gradients, the shell's own gradient rows)
cross xGMI:

  forward   0. every rank compacts the (replicated) Gaussians of its shell into dense arrays
               (lidargs_shell_select: one pass over the means, ~P/N rows survive) -- everything below runs on those
            1. bin them and walk the lists once for transmittance only   -> T_pass_g   [N]
            2. all_gather(T_pass)  (N floats/rank; 0.68 MB at 64x2650) -> T_in_g = prod_{h<g} T_pass_h
            3. composite the shell from T_in_g (the reference's global T < 1e-4 early-out is applied to the
               GLOBAL transmittance, so results match the single-GPU walk)
            4. all_gather([C0, C1, D, T_end, T_hand]_g)  (5 planes/rank) -> image = sum_g partials,
               T_final = T_end of the shell where the walk stopped; also gives each rank what lies
               BEHIND it, which its backward needs (lidargs_shell_compose, one launch)
  backward  5. purely local back-to-front pass per shell, seeded with the behind-sums
            6. per-Gaussian gradients have disjoint support across ranks (a Gaussian is in exactly one shell), so the
               "reduce-scatter" of the packed [P,17] gradient rows is really a permutation: `grad_sync` =
               "reduce_scatter"        each rank ends with rows [r*P/N, (r+1)*P/N): the shell's ~P/N rows go straight to
                                       their index-chunk owners with ONE variable-split all-to-all (68 B x P/N per rank
                                       instead of a dense 68 B x P ring reduce-scatter; the split sizes ride on the
                                       T_pass all-gather and the row index travels as an 18th column, so the exchange
                                       adds no collective of its own)
               "reduce_scatter_dense"  the same result through RCCL reduce_scatter on the dense [P,17] tensor
               "all_reduce"            every rank ends with all rows (dense all-reduce)
               "none"                  every rank keeps only its own shell's rows

The per-rank compute is behind a small backend protocol (select / forward / transmittance / render / compose /
backward) so the collective logic above can be exercised on CPU with gloo (tests inject a CPU backend); the product
backend is HipShellBackend (C ABI: lidargs_shell_select / lidargs_forward_shell / lidargs_render_shell /
lidargs_shell_compose / lidargs_backward_shell).
"""
import contextlib
import ctypes as C
import os
import threading

import torch
import torch.nn as nn

GRAD_WIDTHS = (("means3D", 3), ("means2D", 4), ("colors", 2), ("opacities", 1), ("scales", 3), ("rotations", 4))
GRAD_COLS = sum(w for _, w in GRAD_WIDTHS)  # 17 floats = 68 B per Gaussian
ROW_KEYS = ("means3D", "colors", "opacities", "scales", "rotations")


class TorchDistComm:
    """torch.distributed collectives: backend "nccl" (= RCCL over xGMI on ROCm) or "gloo" on CPU."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(self, t):
        t = t.contiguous()
        flat = torch.empty(self.world * t.numel(), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(flat, t.view(-1), group=self.group)      # concatenated layout (gloo + nccl)
        return flat.view((self.world,) + tuple(t.shape))

    def broadcast(self, t, src=0):
        self.dist.broadcast(t, src=src, group=self.group)
        return t

    def all_reduce(self, t):
        self.dist.all_reduce(t, group=self.group)
        return t

    def all_reduce_async(self, t):
        """Starts the all-reduce and returns a callable that waits for it (lets it overlap the rendering)."""
        work = self.dist.all_reduce(t, group=self.group, async_op=True)
        return work.wait

    def all_reduce_max_async(self, t):
        work = self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group, async_op=True)
        return work.wait

    def all_to_all_rows(self, t, send_counts, recv_counts):
        """Variable-split all-to-all over dim 0: rows [sum(send[:d]), +send[d]) go to rank d."""
        out = torch.empty((int(sum(recv_counts)),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.dist.all_to_all_single(out, t.contiguous(), output_split_sizes=[int(c) for c in recv_counts],
                                    input_split_sizes=[int(c) for c in send_counts], group=self.group)
        return out

    def reduce_scatter_rows(self, t):
        """t: [world*rows, cols] -> this rank's reduced [rows, cols] block."""
        rows = t.shape[0] // self.world
        out = torch.empty((rows, t.shape[1]), dtype=t.dtype, device=t.device)
        if t.is_cuda:
            self.dist.reduce_scatter_tensor(out, t.contiguous(), group=self.group)
        else:  # gloo has no reduce_scatter: all_reduce + slice (CPU tests only)
            full = t.clone()
            self.dist.all_reduce(full, group=self.group)
            out.copy_(full[self.rank * rows:(self.rank + 1) * rows])
        return out


class SingleComm:
    """World of one (no process group): lets the shell path run, and be tested, on a single GPU."""
    rank, world = 0, 1

    def all_gather(self, t):
        return t.unsqueeze(0).clone()

    def broadcast(self, t, src=0):
        return t

    def all_reduce(self, t):
        return t

    def all_reduce_async(self, t):
        return lambda: None

    def all_reduce_max_async(self, t):
        return lambda: None

    def all_to_all_rows(self, t, send_counts, recv_counts):
        return t.clone()

    def reduce_scatter_rows(self, t):
        return t.clone()


def shell_edges(means3D, viewmatrix, world, near, far, bins=2048, scales=None, tile_rad=None, shares=None):
    """Range-shell boundaries: world+1 ascending floats, first = -inf, last = +inf, interior edges = quantiles of
    |p_view| from a histogram over (near, far).

    Unweighted, the quantiles balance the Gaussian COUNT.  The work of a shell is closer to its number of (tile, Gaussian)
    instances, and a near Gaussian covers several times the tiles of a far one; with `scales` [P,3] and `tile_rad` = (tile
    width, tile height) in radians each Gaussian is weighted by a per-Gaussian + per-instance cost estimate from the tiles
    its 3-sigma disc spans, (2a/tw + 1)(2a/th + 1) with a = 3 max(scale)/range, so the shells get thinner towards the
    sensor.  `shares` (world positive numbers, default equal) are the fractions of the total weight the shells should get:
    the handle `rebalance_shares` turns measured per-rank times into.  This is load balancing only: any ascending edges give
    the same image and gradients."""
    V = viewmatrix.reshape(4, 4).to(means3D.dtype)
    p = means3D.detach() @ V[:3, :3] + V[3, :3]
    r = torch.linalg.vector_norm(p, dim=1)
    inside = (r > float(near)) & (r < float(far))
    width = (float(far) - float(near)) / bins
    b = ((r - float(near)) / width).long().clamp_(0, bins - 1)
    if scales is not None and tile_rad is not None:
        a = 3.0 * scales.detach().abs().max(dim=1).values / r.clamp(min=1e-3)
        disc = (2.0 * a / float(tile_rad[0]) + 1.0) * (2.0 * a / float(tile_rad[1]) + 1.0)
        # measured on the 2 M street scene: the binned instances per Gaussian are ~half the disc's tiles (footprint pruning,
        # anisotropy) and saturate near the sensor (beam-fan cull); a frame costs ~0.21 us per Gaussian + ~0.155 us per instance
        w = 1.35 + (0.5 * disc).clamp(max=15.0)
    else:
        w = torch.ones_like(r)
    hist = torch.bincount(b[inside], weights=w[inside].double(), minlength=bins)
    cum = torch.cumsum(hist, 0)
    total = cum[-1].clamp(min=1e-30)
    edges = [float("-inf")]
    if shares is None:
        targets = torch.arange(1, world, device=r.device, dtype=cum.dtype) * (total / world)
    else:
        sh = torch.as_tensor(shares, dtype=cum.dtype, device=r.device).clamp(min=1e-6)
        targets = torch.cumsum(sh / sh.sum(), 0)[:-1] * total
    idx = torch.searchsorted(cum, targets)
    prev = float(near)
    for i in idx.tolist():
        e = max(float(near) + (int(i) + 1) * width, prev + 1e-6)      # strictly ascending even for degenerate histograms
        edges.append(e); prev = e
    edges.append(float("inf"))
    return torch.tensor(edges, dtype=torch.float32, device=means3D.device)


def rebalance_shares(shares, times, damping=0.5, fixed=0.0):
    """One step of measured load balancing: the weight share of shell g is scaled by (mean time / its time), i.e. a slow rank
    gets a thinner shell.  `times` are per-rank compute times of a frame cut with `shares`; `fixed` is the part of a frame that
    does not depend on the shell's size (launch overhead), `damping` in (0, 1] limits the step."""
    import numpy as np
    sh = np.asarray(shares, np.float64)
    t = np.maximum(np.asarray(times, np.float64) - fixed, 1e-3)
    new = sh * (t.mean() / t)
    new = sh + damping * (new / new.sum() - sh / sh.sum())
    new = np.maximum(new, 1e-4)
    return (new / new.sum()).tolist()


_NULL_CTX = contextlib.nullcontext()


_SELECT_FUSED = os.environ.get("LIDARGS_SELECT_FUSED", "0") == "1"      # 1: the one-launch selection (round-6 experiment: bit-identical, no faster -- EXPERIMENTS.md)


class HipShellBackend:
    """Per-rank compute on a HIP device through the C ABI (include/lidargs_rasterizer.h)."""

    def __init__(self):
        from diff_lidargs_rasterization import _C
        self._C = _C
        self.lib = _C._lib
        for name in ("lidargs_wedge_select_count", "lidargs_forward_wedge", "lidargs_backward_wedge", "lidargs_wedge_pack_columns", "lidargs_wedge_unpack_columns",
                     "lidargs_wedge_unpack_grad_rows_add", "lidargs_shell_select", "lidargs_shell_select_count", "lidargs_shell_select_gather", "lidargs_shell_transmittance", "lidargs_shell_compose", "lidargs_shell_pack_grad_rows",
                     "lidargs_shell_unpack_grad_rows", "lidargs_shell_chunk_counts", "lidargs_shell_scatter_radii", "lidargs_shell_select_enqueue",
                     "lidargs_wedge_select_enqueue", "lidargs_forward_shell_enqueue", "lidargs_forward_wedge_enqueue", "lidargs_shell_select_sync",
                     "lidargs_wedge_select_sync", "lidargs_shell_unpack_grad_rows_chunk", "lidargs_shell_pack_grad_rows_live_count",
                     "lidargs_shell_pack_grad_rows_live"):
            getattr(self.lib, name).restype = C.c_int
        self.lib.lidargs_shell_select_scratch_bytes.restype = C.c_size_t
        self._scratch = {}          # persistent scratch of the selection (flags + offsets) per (device, P); never saved for a backward
        self._tls = threading.local()

    fused_chunk_counts = True       # select(..., chunks=(rows, world, out)) leaves the all-to-all's split sizes in `out` (no launch of its own)

    @contextlib.contextmanager
    def frame(self, dev):
        """One device switch and one stream lookup for everything a rank's forward (or backward) enqueues: the per-call
        `torch.cuda.device` / `current_stream` pairs were a fifth of the host's time per frame."""
        if dev.type != "cuda":
            yield
            return
        with torch.cuda.device(dev):
            self._tls.frame = (dev, self._C._stream(dev))
            try:
                yield
            finally:
                self._tls.frame = None

    def _on(self, dev):
        f = getattr(self._tls, "frame", None)
        return _NULL_CTX if (f is not None and f[0] == dev) else torch.cuda.device(dev)

    def _st(self, dev):
        f = getattr(self._tls, "frame", None)
        return f[1] if (f is not None and f[0] == dev) else self._C._stream(dev)

    @staticmethod
    def _chunk_args(chunks):
        if chunks is None:
            return C.c_int(0), C.c_int(0), None
        rows, world, out = chunks
        return C.c_int(int(rows)), C.c_int(int(world)), C.c_void_p(out.data_ptr())

    def _select_enqueue(self, inp, plan, call, chunks=None):
        """Enqueue-only selection into plan.rows rows (no host read): (idx [cap], capacity-row inputs + the device row count)."""
        _C = self._C
        m3 = inp["means3D"]
        _C._require_device(m3, "means3D")
        dev, P, cap = m3.device, int(m3.shape[0]), int(plan.rows)
        key = (dev, P)
        scr = self._scratch.get(key)
        if scr is None:
            nb = int(self.lib.lidargs_shell_select_scratch_bytes(C.c_int(P)))
            scr = (torch.empty(nb, dtype=torch.uint8, device=dev), nb)
            self._scratch = {key: scr}
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        idx = torch.empty(cap, dtype=torch.int32, device=dev)
        sel = dict(inp)
        sel.update(means3D=f(cap, 3), colors=f(cap, 2), opacities=f(cap, 1), scales=f(cap, 3), rotations=f(cap, 4),
                   n_valid=torch.empty(2, dtype=torch.int32, device=dev))
        p = _C._ptr
        with self._on(dev):
            rc = call(p, C.c_int(P), C.c_int(cap), p(idx), p(sel["means3D"]), p(sel["colors"]), p(sel["opacities"]), p(sel["scales"]),
                      p(sel["rotations"]), p(sel["n_valid"]), C.c_void_p(plan.status.data_ptr() + 64), p(scr[0]), C.c_size_t(scr[1]),
                      *self._chunk_args(chunks), self._st(dev))
        if rc < 0:
            _C._raise(rc, "select (enqueue-only)")
        return idx, sel

    def _select_sync(self, inp, call, chunks, what):
        """An ordinary frame's selection in one launch (round 6): P-row arrays per call (the caching allocator hands them out without a
        device malloc; only the first M rows are written and kept as views), one host read for M."""
        _C = self._C
        m3 = inp["means3D"]
        dev, P = m3.device, int(m3.shape[0])
        key = (dev, P)
        scr = self._scratch.get(key)
        if scr is None:
            nb = int(self.lib.lidargs_shell_select_scratch_bytes(C.c_int(P)))
            scr = (torch.empty(nb, dtype=torch.uint8, device=dev), nb)
            self._scratch = {key: scr}
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        idx = torch.empty(P, dtype=torch.int32, device=dev)
        rows = dict(means3D=f(P, 3), colors=f(P, 2), opacities=f(P, 1), scales=f(P, 3), rotations=f(P, 4))
        n_valid = torch.empty(2, dtype=torch.int32, device=dev)
        p = _C._ptr
        with self._on(dev):
            M = call(p, C.c_int(P), C.c_int(P), p(idx), p(rows["means3D"]), p(rows["colors"]), p(rows["opacities"]), p(rows["scales"]), p(rows["rotations"]),
                     p(n_valid), p(scr[0]), C.c_size_t(scr[1]), *self._chunk_args(chunks), self._st(dev))
        if M < 0:
            _C._raise(M, what)
        sel = dict(inp)
        sel.update({k: v[:M] for k, v in rows.items()})
        return idx[:M], sel

    def select(self, inp, lo, hi, plan=None, chunks=None):
        """Step 0: dense copies of the Gaussians with range in [lo, hi) + their indices (ascending).

        The M-row tensors are allocated PER CALL (caching allocator: no device malloc in steady state): a forward's selection
        is saved for its backward, and a second forward before that backward (several views per step, gradient accumulation,
        an eval render) must not overwrite it.  Only the flags/offsets scratch, dead when this returns, is persistent."""
        _C, lib = self._C, self.lib
        m3 = inp["means3D"]
        if plan is not None and int(m3.shape[0]):
            return self._select_enqueue(inp, plan, lambda p, cP, ccap, *rest: lib.lidargs_shell_select_enqueue(
                cP, p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), p(inp["rotations"]), p(inp["viewmatrix"]),
                C.c_float(lo), C.c_float(hi), ccap, *rest), chunks)
        _C._require_device(m3, "means3D")
        dev, P = m3.device, int(m3.shape[0])
        if _SELECT_FUSED and P:
            return self._select_sync(inp, lambda p, cP, *rest: lib.lidargs_shell_select_sync(
                cP, p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), p(inp["rotations"]), p(inp["viewmatrix"]),
                C.c_float(lo), C.c_float(hi), *rest), chunks, "lidargs_shell_select_sync")
        key = (dev, P)
        scr = self._scratch.get(key)
        if scr is None:
            nb = int(lib.lidargs_shell_select_scratch_bytes(C.c_int(P)))
            scr = (torch.empty(nb, dtype=torch.uint8, device=dev), nb)
            self._scratch = {key: scr}
        p = _C._ptr
        M = 0
        if P:
            with self._on(dev):
                M = lib.lidargs_shell_select_count(C.c_int(P), p(m3), p(inp["viewmatrix"]), C.c_float(lo), C.c_float(hi), p(scr[0]),
                                                   C.c_size_t(scr[1]), self._st(dev))
            if M < 0:
                _C._raise(M, "lidargs_shell_select_count")
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        idx = torch.empty(M, dtype=torch.int32, device=dev)
        sel = dict(inp)
        sel.update(means3D=f(M, 3), colors=f(M, 2), opacities=f(M, 1), scales=f(M, 3), rotations=f(M, 4))
        if M:
            with self._on(dev):
                rc = lib.lidargs_shell_select_gather(C.c_int(P), p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]),
                                                     p(inp["rotations"]), p(idx), p(sel["means3D"]), p(sel["colors"]), p(sel["opacities"]),
                                                     p(sel["scales"]), p(sel["rotations"]), p(scr[0]), C.c_size_t(scr[1]),
                                                     *self._chunk_args(chunks), self._st(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_shell_select_gather")
        elif chunks is not None:
            chunks[2].zero_()
        return idx, sel

    def forward(self, inp, lo, hi, plan=None, T_pass=None):
        _C, lib = self._C, self.lib
        m3 = inp["means3D"]
        dev, P, H, W = m3.device, int(m3.shape[0]), inp["H"], inp["W"]
        st = dict(inp=inp, P=P, geom=_C._Scratch(dev), binning=_C._Scratch(dev), img=_C._Scratch(dev))
        st["radii"] = torch.empty(P, dtype=torch.int32, device=dev)          # the library writes every row
        st["radii_xy"] = torch.empty(2 * P, dtype=torch.int32, device=dev)
        if T_pass is None:
            T_pass = torch.empty(H * W, dtype=torch.float32, device=dev)
        if not P:
            T_pass.fill_(1.0)                                              # (with rows, the library writes every pixel)
        dummy = torch.empty(4 * H * W, dtype=torch.float32, device=dev)
        n = 0
        if P:
            p = _C._ptr
            common = (_C._alloc_cb, st["geom"].user, _C._alloc_cb, st["binning"].user, _C._alloc_cb, st["img"].user, C.c_int(P), None,
                      C.c_int(W), C.c_int(H),
                      p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), C.c_float(inp["scale_modifier"]),
                      p(inp["rotations"]), None, p(inp["viewmatrix"]), p(inp["beams"]), C.c_int(inp["far"]), C.c_int(inp["near"]),
                      C.c_float(lo), C.c_float(hi), None, C.c_int(1), p(dummy), p(dummy[2 * H * W:]), p(dummy[3 * H * W:]), p(T_pass),
                      p(st["radii"]), p(st["radii_xy"]), C.c_int(0))
            with self._on(dev):
                if plan is None:
                    n = lib.lidargs_forward_shell(*common, self._st(dev))
                else:       # nothing is read back: capacities from the plan, the row count stays on the device
                    n = lib.lidargs_forward_shell_enqueue(*common, p(inp.get("n_valid")), C.c_int(plan.instances), C.c_int(plan.tile_rows),
                                                          C.c_void_p(plan.status.data_ptr()), self._st(dev))
            if n < 0:
                _C._raise(n, "lidargs_forward_shell")
        for k in ("geom", "binning", "img"):       # keep only the tensors: nothing holds the registry entries alive
            st[k] = st[k].take()
        st["R"] = n
        return st, T_pass

    def transmittance(self, allT, rank):
        """Step 2: T_in = product of the hand-over transmittances of the shells in front.  allT: [G, N]."""
        G, N = int(allT.shape[0]), int(allT.shape[1])
        T_in = torch.empty(N, dtype=torch.float32, device=allT.device)
        if allT.stride(1) != 1 or (G > 1 and allT.stride(0) < N):
            allT = allT.contiguous()
        stride = int(allT.stride(0)) if G > 1 else N                       # (the gathered rows may carry the split sizes behind their N values)
        with self._on(allT.device):
            rc = self.lib.lidargs_shell_transmittance(C.c_int(G), C.c_int(rank), C.c_int(N), C.c_size_t(stride), C.c_void_p(allT.data_ptr()),
                                                      self._C._ptr(T_in), self._st(allT.device))
        if rc < 0:
            self._C._raise(rc, "lidargs_shell_transmittance")
        return T_in

    def render(self, st, T_in):
        """Step 3 -> planes [5, N]: colour0, colour1, depth partial sums, T_end, T_hand."""
        _C, lib = self._C, self.lib
        inp = st["inp"]
        dev, H, W = T_in.device, inp["H"], inp["W"]
        N = H * W
        planes = torch.empty(6 * N, dtype=torch.float32, device=dev)     # [C0, C1, D, T_end, T_hand, occ scratch]
        if st["P"]:
            p = _C._ptr
            with self._on(dev):
                rc = lib.lidargs_render_shell(C.c_int(st["P"]), C.c_int(st["R"]), None, C.c_int(W), C.c_int(H), p(st["geom"]),
                                              p(st["binning"]), p(st["img"]), p(T_in.contiguous()), C.c_int(0), p(planes),
                                              p(planes[2 * N:]), p(planes[5 * N:]), p(planes[4 * N:]), p(planes[3 * N:]), C.c_int(0),
                                              self._st(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_render_shell")
        else:
            planes[:3 * N] = 0
            planes[3 * N:4 * N] = T_in
            planes[4 * N:5 * N] = T_in
        return planes[:5 * N].view(5, N)

    def compose(self, planes, rank, bg, H, W):
        """Step 4: planes [G, 5, N] -> (color [2,H,W], depth [1,H,W], occ [1,H,W], T_final [N], behind [3,N])."""
        _C = self._C
        G, N, dev = int(planes.shape[0]), H * W, planes.device
        out = torch.empty(8 * N, dtype=torch.float32, device=dev)
        color, depth, occ, T_final, behind = out[:2 * N], out[2 * N:3 * N], out[3 * N:4 * N], out[4 * N:5 * N], out[5 * N:]
        p = _C._ptr
        with self._on(dev):
            rc = self.lib.lidargs_shell_compose(C.c_int(G), C.c_int(rank), C.c_int(N), p(planes.contiguous()), p(bg), p(color), p(depth), p(occ),
                                                p(T_final), p(behind), self._st(dev))
        if rc < 0:
            _C._raise(rc, "lidargs_shell_compose")
        return color.view(2, H, W), depth.view(1, H, W), occ.view(1, H, W), T_final, behind.view(3, N)

    def backward(self, st, behind, T_final, grads):
        _C, lib = self._C, self.lib
        inp = st["inp"]
        P, H, W = st["P"], inp["H"], inp["W"]
        dev = behind.device
        widths = (3, 4, 2, 1, 3, 4, 6)                             # the six returned gradients + dL/dcov3D
        slab = torch.empty(P * sum(widths), dtype=torch.float32, device=dev)      # the library writes every row
        parts, o = [], 0
        for w in widths:
            parts.append(slab[o:o + P * w].view(P, w)); o += P * w
        (g_m3, g_m2, g_col, g_op, g_sc, g_rot, g_cov) = parts
        g_con = g_dep = g_sph = g_u1 = g_u2 = None                 # the reference's scratch gradients: not materialised
        if P:
            p = _C._ptr
            gc, gd, go = (g.contiguous() for g in grads)
            with self._on(dev):
                rc = lib.lidargs_backward_shell(
                    C.c_int(P), C.c_int(st["R"]), p(inp["bg"]), C.c_int(W), C.c_int(H), p(inp["means3D"]), p(inp["colors"]), p(inp["scales"]),
                    C.c_float(inp["scale_modifier"]), p(inp["rotations"]), None, p(inp["viewmatrix"]), p(inp["beams"]), p(st["radii"]),
                    p(st["geom"]), p(st["binning"]), p(st["img"]), p(behind.contiguous()), p(T_final.contiguous()),
                    p(gc), p(gd), p(go), p(g_m2), p(g_con), p(g_op), p(g_col), p(g_dep), p(g_m3), p(g_sph), p(g_u1), p(g_u2), p(g_cov),
                    p(g_sc), p(g_rot), C.c_int(0), self._st(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_backward_shell")
        return dict(means3D=g_m3, means2D=g_m2, colors=g_col, opacities=g_op, scales=g_sc, rotations=g_rot)


    # ---- column wedges -------------------------------------------------------------------------------------------------
    def select_wedge(self, inp, c0, c1, plan=None, chunks=None):
        """Dense copies of the Gaussians whose rect can reach pixel columns [c0, c1) + their indices (ascending); M rows per call."""
        _C, lib = self._C, self.lib
        m3 = inp["means3D"]
        if plan is not None and int(m3.shape[0]):
            return self._select_enqueue(inp, plan, lambda p, cP, ccap, *rest: lib.lidargs_wedge_select_enqueue(
                cP, p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), p(inp["rotations"]), C.c_float(inp["scale_modifier"]),
                p(inp["viewmatrix"]), C.c_int(inp["W"]), C.c_int(c0), C.c_int(c1), ccap, *rest), chunks)
        _C._require_device(m3, "means3D")
        dev, P = m3.device, int(m3.shape[0])
        if _SELECT_FUSED and P:
            return self._select_sync(inp, lambda p, cP, *rest: lib.lidargs_wedge_select_sync(
                cP, p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), p(inp["rotations"]), C.c_float(inp["scale_modifier"]),
                p(inp["viewmatrix"]), C.c_int(inp["W"]), C.c_int(c0), C.c_int(c1), *rest), chunks, "lidargs_wedge_select_sync")
        key = (dev, P)
        scr = self._scratch.get(key)
        if scr is None:
            nb = int(lib.lidargs_shell_select_scratch_bytes(C.c_int(P)))
            scr = (torch.empty(nb, dtype=torch.uint8, device=dev), nb)
            self._scratch = {key: scr}
        p = _C._ptr
        M = 0
        if P:
            with self._on(dev):
                M = lib.lidargs_wedge_select_count(C.c_int(P), p(m3), p(inp["scales"]), p(inp["rotations"]), C.c_float(inp["scale_modifier"]),
                                                   p(inp["viewmatrix"]), C.c_int(inp["W"]), C.c_int(c0), C.c_int(c1), p(scr[0]), C.c_size_t(scr[1]),
                                                   self._st(dev))
            if M < 0:
                _C._raise(M, "lidargs_wedge_select_count")
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
        idx = torch.empty(M, dtype=torch.int32, device=dev)
        sel = dict(inp)
        sel.update(means3D=f(M, 3), colors=f(M, 2), opacities=f(M, 1), scales=f(M, 3), rotations=f(M, 4))
        if M:
            with self._on(dev):
                rc = lib.lidargs_shell_select_gather(C.c_int(P), p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]),
                                                     p(inp["rotations"]), p(idx), p(sel["means3D"]), p(sel["colors"]), p(sel["opacities"]),
                                                     p(sel["scales"]), p(sel["rotations"]), p(scr[0]), C.c_size_t(scr[1]),
                                                     *self._chunk_args(chunks), self._st(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_shell_select_gather")
        elif chunks is not None:
            chunks[2].zero_()
        return idx, sel

    def forward_wedge(self, inp, c0, c1, plan=None):
        """lidargs_forward_wedge on the selected rows -> state for the backward, planes [4, H, W] (colour 0/1, depth, occupancy;
        only columns [c0, c1) are this rank's)."""
        _C, lib = self._C, self.lib
        m3 = inp["means3D"]
        dev, P, H, W = m3.device, int(m3.shape[0]), inp["H"], inp["W"]
        st = dict(inp=inp, P=P, geom=_C._Scratch(dev), binning=_C._Scratch(dev), img=_C._Scratch(dev))
        st["radii"] = torch.empty(P, dtype=torch.int32, device=dev)
        st["radii_xy"] = torch.empty(2 * P, dtype=torch.int32, device=dev)
        planes = torch.empty(4 * H * W, dtype=torch.float32, device=dev)
        n = 0
        if P:
            p = _C._ptr
            N = H * W
            common = (_C._alloc_cb, st["geom"].user, _C._alloc_cb, st["binning"].user, _C._alloc_cb, st["img"].user, C.c_int(P), p(inp["bg"]),
                      C.c_int(W), C.c_int(H), p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), C.c_float(inp["scale_modifier"]),
                      p(inp["rotations"]), None, p(inp["viewmatrix"]), p(inp["beams"]), C.c_int(inp["far"]), C.c_int(inp["near"]),
                      C.c_int(c0), C.c_int(c1), p(planes), p(planes[2 * N:]), p(planes[3 * N:]), p(st["radii"]), p(st["radii_xy"]), C.c_int(0))
            with self._on(dev):
                if plan is None:
                    n = lib.lidargs_forward_wedge(*common, self._st(dev))
                else:
                    n = lib.lidargs_forward_wedge_enqueue(*common, p(inp.get("n_valid")), C.c_int(plan.instances), C.c_int(plan.tile_rows),
                                                          C.c_void_p(plan.status.data_ptr()), self._st(dev))
            if n < 0:
                _C._raise(n, "lidargs_forward_wedge")
        else:       # no Gaussian can reach the wedge: background only
            pl = planes.view(4, H * W)
            pl[0] = inp["bg"][0]; pl[1] = inp["bg"][1]; pl[2] = 0; pl[3] = 0
        for k in ("geom", "binning", "img"):
            st[k] = st[k].take()
        st["R"] = n
        st["cols"] = (c0, c1)
        return st, planes

    def backward_plain(self, st, grads):
        """lidargs_backward_wedge on a wedge's forward state; grads = (colour [2,N], depth [N], occ [N]), full-size planes."""
        _C, lib = self._C, self.lib
        inp = st["inp"]
        P, H, W = st["P"], inp["H"], inp["W"]
        dev = grads[0].device
        widths = (3, 4, 2, 1, 3, 4, 6)
        slab = torch.empty(P * sum(widths), dtype=torch.float32, device=dev)
        parts, o = [], 0
        for w in widths:
            parts.append(slab[o:o + P * w].view(P, w)); o += P * w
        (g_m3, g_m2, g_col, g_op, g_sc, g_rot, g_cov) = parts
        if P:
            p = _C._ptr
            gc, gd, go = (g.contiguous() for g in grads)
            with self._on(dev):
                rc = lib.lidargs_backward_wedge(
                    C.c_int(P), C.c_int(st["R"]), p(inp["bg"]), C.c_int(W), C.c_int(H), p(inp["means3D"]), p(inp["colors"]), p(inp["scales"]),
                    C.c_float(inp["scale_modifier"]), p(inp["rotations"]), None, p(inp["viewmatrix"]), p(inp["beams"]), p(st["radii"]),
                    p(st["geom"]), p(st["binning"]), p(st["img"]), C.c_int(st["cols"][0]), C.c_int(st["cols"][1]), p(gc), p(gd), p(go),
                    p(g_m2), p(g_op), p(g_col), p(g_m3), p(g_cov), p(g_sc), p(g_rot), C.c_int(0), self._st(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_backward_wedge")
        return dict(means3D=g_m3, means2D=g_m2, colors=g_col, opacities=g_op, scales=g_sc, rotations=g_rot)

    def pack_columns(self, planes, H, W, c0, c1, wmax, out):
        """out f32[4*H*wmax (+ tail)]: this rank's columns of the four planes, zero padded to wmax columns."""
        p = self._C._ptr
        N = H * W
        self._call("lidargs_wedge_pack_columns", planes.device, C.c_int(H), C.c_int(W), C.c_int(c0), C.c_int(c1), C.c_int(wmax), p(planes),
                   p(planes[2 * N:]), p(planes[3 * N:]), p(out))

    def unpack_columns(self, blocks, edges, H, W, wmax):
        """blocks [G, stride] gathered from the ranks -> (color [2,H,W], depth [1,H,W], occ [1,H,W])."""
        p, dev = self._C._ptr, blocks.device
        G, stride = int(blocks.shape[0]), int(blocks.shape[1])
        out = torch.empty(4 * H * W, dtype=torch.float32, device=dev)
        N = H * W
        e = (C.c_int * (G + 1))(*[int(x) for x in edges])
        self._call("lidargs_wedge_unpack_columns", dev, C.c_int(G), C.c_int(H), C.c_int(W), C.c_int(wmax), C.c_size_t(stride), e,
                   p(blocks.contiguous()), p(out), p(out[2 * N:]), p(out[3 * N:]))
        return out[:2 * N].view(2, H, W), out[2 * N:3 * N].view(1, H, W), out[3 * N:].view(1, H, W)

    def unpack_rows_add(self, rows, P):
        """Flat [17 P] tensor of six contiguous gradient blocks; rows carrying the same index are added."""
        dev, p = rows.device, self._C._ptr
        rows = rows.contiguous()
        dense = torch.empty(P * GRAD_COLS, dtype=torch.float32, device=dev)
        self._call("lidargs_wedge_unpack_grad_rows_add", dev, C.c_int(int(rows.shape[0])), p(rows), C.c_int(P), p(dense))
        return dense

    # ---- step 6 helpers: one launch each instead of concatenates, casts and index copies -------------------------------
    def unpack_rows_chunk(self, rows, base, chunk_rows, add=False):
        """Gradient mode "shard": [n, 18] rows -> this rank's own [17 * chunk_rows] block (six contiguous gradient blocks)."""
        dev = rows.device
        dense = torch.empty(GRAD_COLS * int(chunk_rows), dtype=torch.float32, device=dev)
        self._call("lidargs_shell_unpack_grad_rows_chunk", dev, C.c_int(int(rows.shape[0])), self._C._ptr(rows), C.c_int(int(base)), C.c_int(int(chunk_rows)),
                   self._C._ptr(dense) if chunk_rows else None, C.c_int(1 if add else 0))
        return dense

    def _call(self, name, dev, *args):
        with self._on(dev):
            rc = getattr(self.lib, name)(*args, self._st(dev))
        if rc < 0:
            self._C._raise(rc, name)

    def scatter_radii(self, idx, radii_shell, P):
        """radii i32[P]: the shell's radii at their global rows, zero elsewhere."""
        dev, p = idx.device, self._C._ptr
        out = torch.empty(P, dtype=torch.int32, device=dev)
        self._call("lidargs_shell_scatter_radii", dev, C.c_int(int(idx.shape[0])), p(idx), p(radii_shell), C.c_int(P), p(out))
        return out

    def chunk_counts(self, idx, chunk_rows, world, out):
        """out f32[world] (a view into the buffer the T_pass all-gather ships): rows of this shell bound for each index chunk."""
        self._call("lidargs_shell_chunk_counts", idx.device, C.c_int(int(idx.shape[0])), self._C._ptr(idx), C.c_int(chunk_rows), C.c_int(world),
                   self._C._ptr(out))

    def pack_rows(self, g, idx):
        """[M, 18]: the six gradients of the shell's rows + the bit pattern of their global index."""
        dev, p, M = idx.device, self._C._ptr, int(idx.shape[0])
        rows = torch.empty((M, GRAD_COLS + 1), dtype=torch.float32, device=dev)
        self._call("lidargs_shell_pack_grad_rows", dev, C.c_int(M), p(g["means3D"]), p(g["means2D"]), p(g["colors"]), p(g["opacities"]),
                   p(g["scales"]), p(g["rotations"]), p(idx), p(rows))
        return rows

    def _live_args(self, g, idx, P, chunk_rows, world):
        p = self._C._ptr
        return (C.c_int(int(idx.shape[0])), p(g["means3D"]), p(g["means2D"]), p(g["colors"]), p(g["opacities"]), p(g["scales"]), p(g["rotations"]), p(idx),
                C.c_int(int(P)), C.c_int(int(chunk_rows)), C.c_int(int(world)))

    def count_rows_live(self, g, idx, P, chunk_rows, world):
        """Rows that carry a gradient, per destination chunk (round 6): float32 [world] ON THE DEVICE, no host read (exact: < 2^24 rows per chunk)."""
        dev = idx.device
        cnt = torch.empty(2 * world, dtype=torch.int32, device=dev)
        with self._on(dev):
            rc = self.lib.lidargs_shell_pack_grad_rows_live_count(*self._live_args(g, idx, P, chunk_rows, world), C.c_void_p(cnt.data_ptr()), None, self._st(dev))
        if rc < 0:
            self._C._raise(rc, "lidargs_shell_pack_grad_rows_live_count")
        return cnt

    def pack_rows_live(self, g, idx, P, chunk_rows, world, cnt, n):
        """The n = sum(counts) live rows, [n, 18], grouped by destination chunk; `cnt` = what count_rows_live returned."""
        dev = idx.device
        rows = torch.empty((int(n), GRAD_COLS + 1), dtype=torch.float32, device=dev)
        if n:
            with self._on(dev):
                rc = self.lib.lidargs_shell_pack_grad_rows_live(*self._live_args(g, idx, P, chunk_rows, world), C.c_void_p(cnt.data_ptr()),
                                                                C.c_void_p(cnt.data_ptr() + 4 * world), self._C._ptr(rows), self._st(dev))
            if rc < 0:
                self._C._raise(rc, "lidargs_shell_pack_grad_rows_live")
        return rows

    def unpack_rows(self, rows, P, blocked=False):
        """Zero, then every row written at the index it carries: dense [P, 17], or (blocked) one flat [17 P] tensor holding the
        six gradients as contiguous blocks [P,3][P,4][P,2][P,1][P,3][P,4], which autograd takes without a strided copy each."""
        dev, p = rows.device, self._C._ptr
        rows = rows.contiguous()
        dense = torch.empty(P * GRAD_COLS if blocked else (P, GRAD_COLS), dtype=torch.float32, device=dev)
        self._call("lidargs_shell_unpack_grad_rows", dev, C.c_int(int(rows.shape[0])), p(rows), C.c_int(P), p(dense), C.c_int(1 if blocked else 0))
        return dense



class _RankPlan:
    """Caller-side state of ENQUEUE-ONLY rank frames (`module.enqueue_only`, or LIDARGS_ENQUEUE_ONLY=1): a rank's frame has two host
    reads -- the number of selected rows, the number of list instances -- and each of them lets the device run dry while the host
    catches up.  The first frame runs that way and teaches the plan both numbers; every later frame only enqueues work into
    capacities with headroom (lidargs_*_select_enqueue, lidargs_forward_*_enqueue): the counts stay on the device and 18 status words
    come back through pinned memory.  A frame that needed more than its capacities is found out by the backward of that frame (it
    waits for the forward's status anyway, for the all-to-all's split sizes) or by the next forward: RuntimeError, capacities raised."""
    HEADROOM = 1.25

    def __init__(self):
        self.rows = self.instances = 0
        self.tile_rows = 4
        self.status = None                # pinned int32[18]: [0..15] lidargs_forward_*_enqueue's words, [16] rows gathered, [17] rows selected
        self.event, self.pending = None, False
        self.frames = 0

    def learn(self, rows, num_rendered):
        need = int(num_rendered) & ~3
        self.tile_rows = 4 << (int(num_rendered) & 3)
        self.rows = max(self.rows, int(rows * self.HEADROOM) + 256)
        self.instances = max(self.instances, (int(need * self.HEADROOM) + 4096 + 3) & ~3)

    def next(self):
        """The plan for an enqueue-only frame, or None while nothing has been learnt (an ordinary frame, which teaches it)."""
        if not self.rows:
            return None
        # the status words and the event are shared by every frame in flight: the previous frame's words are read (one event wait, on
        # a frame that is already a frame old) BEFORE this frame's copy may overwrite them -- a look that returns when the event has not
        # completed yet, which is the normal state with the host running ahead, would lose that frame's overflow flag
        self.check(wait=True)
        if self.status is None:
            self.status = torch.zeros(18, dtype=torch.int32).pin_memory()
        return self

    def submitted(self):
        if self.event is None:
            self.event = torch.cuda.Event()
        self.event.record()
        self.pending = True
        self.frames += 1

    def check(self, wait=True):
        if not self.pending or (not wait and not self.event.query()):
            return
        self.event.synchronize()
        self.pending = False
        st = self.status.tolist()
        need, over, selected = st[0], st[8], st[17]
        rows_over = selected > self.rows
        msg = None
        if over:
            msg = f"needed {need} list instances but its binning buffer held {st[9]}"
        if rows_over:
            msg = f"selected {selected} Gaussians but had room for {self.rows}"
        if need * 1.08 > self.instances:
            self.instances = (int(need * self.HEADROOM) + 4096 + 3) & ~3
        if selected * 1.08 > self.rows:
            self.rows = int(selected * self.HEADROOM) + 256
        if msg:
            raise RuntimeError(f"lidargs_dist: an enqueue-only rank frame {msg}; that frame's outputs are invalid (capacities raised, re-render it)")


class _HostCounts:
    """The gradient all-to-all's split sizes [src, dst], on their way to the host: the device-to-host copy is queued on the stream (pinned
    memory) behind the collective that delivered them and an event is recorded; the host waits for the event only where it needs the
    numbers -- in the backward, in front of the all-to-all, with the backward's own kernels already queued -- instead of draining the
    stream at the end of the forward (`.cpu()`: one full host / device serialisation per frame less)."""

    def __init__(self, counts):
        if counts.is_cuda:
            self.host = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)     # (as shipped: exact floats, converted on the host)
            self.host.copy_(counts, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = counts, None

    def splits(self, rank):
        if self.event is not None:
            self.event.synchronize()
            self.event = None
        return [int(v) for v in self.host[rank].tolist()], [int(v) for v in self.host[:, rank].tolist()]


def _chunk_rows(P, world):
    return (P + world - 1) // world


def _shell_forward(module, means3D, colors, opacities, scales, rotations):
    """Steps 0-4 of the module docstring.  Returns ((color, depth, occ, radii), saved-for-backward)."""
    rs, comm, be = module.raster_settings, module.comm, module.backend
    H, W = int(rs.image_height), int(rs.image_width)
    N = H * W
    dev = means3D.device
    P = int(means3D.shape[0])
    f32 = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous()
    inp = dict(means3D=f32(means3D), colors=f32(colors), opacities=f32(opacities), scales=f32(scales), rotations=f32(rotations),
               viewmatrix=f32(rs.viewmatrix), beams=f32(rs.beam_inclinations), H=H, W=W, scale_modifier=float(rs.scale_modifier),
               far=int(rs.lidar_far), near=int(rs.lidar_near), bg=rs.bg.to(torch.float32).to(dev).contiguous())
    edges = module.edges
    if edges is None:
        edges = shell_edges(inp["means3D"], inp["viewmatrix"], comm.world, rs.lidar_near, rs.lidar_far)
        edges = comm.broadcast(edges, 0)       # every rank must cut at the same ranges
    if not isinstance(edges, (list, tuple)):
        edges = [float(e) for e in edges.tolist()]
        if module.edges is not None:
            module.edges = edges               # static cut: convert once, no device read per frame
    lo, hi = edges[comm.rank], edges[comm.rank + 1]

    exchange = comm.world > 1 and module.grad_sync in ("reduce_scatter", "shard")
    fused = getattr(be, "fused_chunk_counts", False)                              # (the HIP backend; the framework-op backend of the CPU tests is not)
    enqueue_only = module.enqueue_only and fused
    plan = module.plan.next() if enqueue_only else None                           # None: an ordinary frame (two host reads)
    tail = ship = None
    if exchange:
        # split sizes of the gradient all-to-all: the selection is index-sorted, so the rows bound for index chunk d are
        # contiguous.  They ride on the T_pass all-gather (exact as floats: < 2^24 rows per chunk) instead of a collective
        # of their own, and are read back at the end of the forward, off the backward's critical path.
        rows = _chunk_rows(P, comm.world)
        assert rows < (1 << 24)
        ship = torch.empty(N + comm.world, dtype=torch.float32, device=dev)      # what the all-gather ships: T_pass, then the split sizes
        tail = ship[N:]
    if fused:       # (the selection's own gather launch leaves the split sizes in `tail`, the forward writes T_pass in place)
        idx, sel = be.select(inp, lo, hi, plan, chunks=(rows, comm.world, tail) if exchange else None)   # 0   [M], M-row inputs
    else:
        idx, sel = be.select(inp, lo, hi)
        if exchange:
            be.chunk_counts(idx, rows, comm.world, tail)
    # `sel` already holds exactly this shell's rows: the shell test is NOT repeated inside the forward (two kernels need not
    # round the same range expression identically; a Gaussian one ulp from an edge could be selected here and culled there)
    if fused:
        st, T_pass = be.forward(sel, float("-inf"), float("inf"), plan, T_pass=None if ship is None else ship[:N])   # 1
    else:
        st, T_pass = be.forward(sel, float("-inf"), float("inf"))
        if exchange:
            ship[:N] = T_pass
    radii = be.scatter_radii(idx, st["radii"], P)
    wait_radii = comm.all_reduce_async(radii) if comm.world > 1 else (lambda: None)   # overlaps the rendering
    allT = comm.all_gather(T_pass if ship is None else ship)                       # 2   [G, N (+G)]
    counts = allT[:, N:] if exchange else None
    T_in = be.transmittance(allT[:, :N] if exchange else allT, comm.rank)
    planes = comm.all_gather(be.render(st, T_in))                                 # 3, 4   [G, 5, N]
    color, depth, occ, T_final, behind = be.compose(planes, comm.rank, inp["bg"], H, W)
    saved = dict(st=st, behind=behind, T_final=T_final, idx=idx, P=P)
    if exchange:
        saved.update(counts=_HostCounts(counts))                                  # [src, dst], read in the backward
    if enqueue_only:
        if plan is None:
            module.plan.learn(int(idx.shape[0]), st["R"])
        else:
            module.plan.submitted()
    wait_radii()
    return (color, depth, occ, radii), saved


def _shell_backward(module, saved, g_color, g_depth, g_occ):
    """Steps 5-6.  Returns {means3D, means2D, colors, opacities, scales, rotations} gradients, dense [P, w]."""
    st, idx, P = saved["st"], saved["idx"], saved["P"]
    comm, be = module.comm, module.backend
    inp = st["inp"]
    H, W = inp["H"], inp["W"]
    dev = g_color.device
    # d(color)/d(T_final) for the background is inside the blend: (-T_final/(1-alpha)) * bg.g  (R3/cr/backward.cu:727)
    g = be.backward(st, saved["behind"], saved["T_final"], (g_color.reshape(2, H * W), g_depth.reshape(H * W), g_occ.reshape(H * W)))
    sync = module.grad_sync if comm.world > 1 else "none"
    blocked = sync != "reduce_scatter_dense"
    if module.grad_sync == "shard":
        return _shard_grads(module, saved, g, idx, P, add=False)
    if sync == "reduce_scatter":
        # 6: the shell's rows go straight to their index-chunk owners; the row index travels as an 18th column (bit pattern)
        got = _exchange_rows(module, saved, g, idx, P)                  # (every sync mode: a frame over its capacities raises in its own backward, plan.check())
        dense = be.unpack_rows(got, P, blocked=True)
    else:
        packed = be.pack_rows(g, idx)                                                 # [M, 18]: gradients + the row's global index
        module.plan.check()
        dense = be.unpack_rows(packed, P, blocked=blocked)
        if sync == "all_reduce":
            dense = comm.all_reduce(dense)
        elif sync == "reduce_scatter_dense":
            rows = _chunk_rows(P, comm.world)
            pad = rows * comm.world - P
            if pad:
                dense = torch.cat([dense, dense.new_zeros(pad, GRAD_COLS)], 0)
            mine = comm.reduce_scatter_rows(dense)
            dense = dense.new_zeros(rows * comm.world, GRAD_COLS)
            dense[comm.rank * rows:(comm.rank + 1) * rows] = mine
            dense = dense[:P]
    o, out = 0, {}
    for k, w in GRAD_WIDTHS:
        if blocked:
            out[k] = dense[o * P:(o + w) * P].view(P, w)
        else:
            out[k] = dense[:, o:o + w]
        o += w
    return out


SHIP_LIVE_BYTES = 32 << 20      # "auto": live rows only when a rank's share of the full exchange (72 B x P / world) is at least this


def _ship_live_default():
    e = os.environ.get("LIDARGS_SHIP_LIVE", "auto")
    return True if e == "1" else (False if e == "0" else "auto")


def _ships_live(module, P):
    """The same answer on every rank (it depends on P and the world size only: the live form has a collective of its own).  Shipping only the
    live rows costs one host read in the backward and a 4 x world-byte all-gather; it pays when the full exchange is large: 8 M Gaussians over
    8 ranks ship 84 MB per rank otherwise (0.1 MB live), 2 M ship 18 MB (0.6 MB live) -- there the read costs what the bytes save."""
    live = getattr(module, "ship_live", False)
    if live == "auto":
        live = 72 * P // max(1, module.comm.world) >= SHIP_LIVE_BYTES
    return bool(live) and not module.enqueue_only and hasattr(module.backend, "pack_rows_live")


def _exchange_rows(module, saved, g, idx, P):
    """The rows this rank receives for its index chunk: [n, 18] (17 gradient columns + the bit pattern of the row's global index).
    Round 6 (module.ship_live; "auto" = for large exchanges, _ships_live): only the rows that carry a gradient travel -- each rank counts them per destination, the counts
    cross in one small all-gather, then the variable-split all-to-all ships exactly those (cfg4, world 8: 84 MB per rank and frame -> 0.1 MB;
    cfg3: a fifth).  Otherwise (enqueue-only frames: no host read): every selected row, split sizes from the forward's selection."""
    comm, be = module.comm, module.backend
    if _ships_live(module, P):
        chunk = _chunk_rows(P, comm.world)
        cnt = be.count_rows_live(g, idx, P, chunk, comm.world)          # on the device
        module.plan.check()
        allc = comm.all_gather(cnt[:comm.world])                         # [src, dst]: every rank's counts, then ONE host read for send and receive sizes
        host = allc.tolist()
        send, recv = [int(v) for v in host[comm.rank]], [int(row[comm.rank]) for row in host]
        return comm.all_to_all_rows(be.pack_rows_live(g, idx, P, chunk, comm.world, cnt, sum(send)), send, recv)
    packed = be.pack_rows(g, idx)
    module.plan.check()
    send, recv = saved["counts"].splits(comm.rank)
    return comm.all_to_all_rows(packed[:sum(send)], send, recv)


def shard_rows(P, world, rank):
    """(first row, number of rows) of rank's index chunk: the rows whose gradients it receives under grad_sync "reduce_scatter" / "shard"."""
    rows = _chunk_rows(P, world)
    lo = min(P, rank * rows)
    return lo, min(P, lo + rows) - lo


def _shard_grads(module, saved, g, idx, P, add):
    """grad_sync = "shard" (round 6): the rank's own chunk of every gradient, [rows_r, w] each -- what a rank that optimises only its shard
    of the (replicated) Gaussians needs.  The rows travel as under "reduce_scatter" (one variable-split all-to-all); they are unpacked into
    a [17, rows_r] block instead of a zero-filled dense [P, 17] one."""
    comm, be = module.comm, module.backend
    base, n = shard_rows(P, comm.world, comm.rank)
    if comm.world > 1:
        got = _exchange_rows(module, saved, g, idx, P)
    else:
        got = be.pack_rows(g, idx)
        module.plan.check()
    dense = be.unpack_rows_chunk(got, base, n, add=add)
    o, out = 0, {}
    for k, w in GRAD_WIDTHS:
        out[k] = dense[o * n:(o + w) * n].view(n, w)
        o += w
    return out


def _frame_of(be, dev):
    f = getattr(be, "frame", None)
    return f(dev) if f is not None else _NULL_CTX


def shell_forward(module, means3D, colors, opacities, scales, rotations):
    with _frame_of(module.backend, means3D.device):
        return _shell_forward(module, means3D, colors, opacities, scales, rotations)


def shell_backward(module, saved, g_color, g_depth, g_occ):
    with _frame_of(module.backend, g_color.device):
        return _shell_backward(module, saved, g_color, g_depth, g_occ)


def wedge_forward(module, means3D, colors, opacities, scales, rotations):
    with _frame_of(module.backend, means3D.device):
        return _wedge_forward(module, means3D, colors, opacities, scales, rotations)


def wedge_backward(module, saved, g_color, g_depth, g_occ):
    with _frame_of(module.backend, g_color.device):
        return _wedge_backward(module, saved, g_color, g_depth, g_occ)


class _ShellRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, module):
        outs, saved = shell_forward(module, means3D, colors, opacities, scales, rotations)
        ctx.module, ctx.saved = module, saved
        ctx.mark_non_differentiable(outs[3])
        return outs

    @staticmethod
    def backward(ctx, g_color, g_depth, g_occ, _g_radii):
        g = shell_backward(ctx.module, ctx.saved, g_color, g_depth, g_occ)
        return g["means3D"], g["means2D"], g["colors"], g["opacities"], g["scales"], g["rotations"], None


class _ShardRasterize(torch.autograd.Function):
    """grad_sync = "shard": the frame is rendered from the REPLICATED tensors (no gradient flows to them); the gradients come back for the
    rank's own shard leaves [rows_r, w] -- rows [r * rows, r * rows + rows_r) of the replicated ones, which the caller keeps equal to them."""

    @staticmethod
    def forward(ctx, means3D, colors, opacities, scales, rotations, s_means3D, s_means2D, s_colors, s_opacities, s_scales, s_rotations, module, wedges):
        outs, saved = (wedge_forward if wedges else shell_forward)(module, means3D, colors, opacities, scales, rotations)
        ctx.module, ctx.saved, ctx.wedges = module, saved, wedges
        ctx.mark_non_differentiable(outs[3])
        return outs

    @staticmethod
    def backward(ctx, g_color, g_depth, g_occ, _g_radii):
        g = (wedge_backward if ctx.wedges else shell_backward)(ctx.module, ctx.saved, g_color, g_depth, g_occ)
        return (None, None, None, None, None, g["means3D"], g["means2D"], g["colors"], g["opacities"], g["scales"], g["rotations"], None, None)


def _shard_apply(module, wedges, means3D, opacities, colors, scales, rotations, shard):
    P = int(means3D.shape[0])
    base, n = shard_rows(P, module.comm.world, module.comm.rank)
    keys = ("means3D", "means2D", "colors", "opacities", "scales", "rotations")
    if shard is None or any(k not in shard for k in keys):
        raise ValueError('grad_sync="shard": pass shard=dict(means3D, means2D, colors, opacities, scales, rotations) -- this rank\'s own rows as leaves')
    for k, w in GRAD_WIDTHS:
        if tuple(shard[k].shape) != (n, w):
            raise ValueError(f'grad_sync="shard": shard["{k}"] must be [{n}, {w}] (rows {base}..{base + n} of the replicated tensor), got {tuple(shard[k].shape)}')
    d = lambda t: t.detach()
    return _ShardRasterize.apply(d(means3D), d(colors), d(opacities), d(scales), d(rotations), shard["means3D"], shard["means2D"], shard["colors"],
                                 shard["opacities"], shard["scales"], shard["rotations"], module, wedges)


class ShellRasterizer(nn.Module):
    """Range-shell sharded counterpart of GaussianRasterizer.forward (colors_precomp + scales/rotations path,
    the one gaussian_renderer.render() uses).  Inputs are REPLICATED on every rank; outputs are identical
    on every rank; gradients follow `grad_sync`."""

    def __init__(self, raster_settings, comm=None, backend=None, grad_sync="reduce_scatter", edges=None):
        super().__init__()
        assert grad_sync in ("reduce_scatter", "reduce_scatter_dense", "all_reduce", "none", "shard")
        self.raster_settings = raster_settings
        self.comm = comm if comm is not None else SingleComm()
        self.backend = backend if backend is not None else HipShellBackend()
        self.grad_sync = grad_sync
        self.edges = edges
        self.enqueue_only = os.environ.get("LIDARGS_ENQUEUE_ONLY", "0") == "1"     # see _RankPlan; off by default
        self.ship_live = _ship_live_default()      # the gradient exchange ships only rows with a gradient: True / False / "auto" (_exchange_rows)
        self.plan = _RankPlan()

    def forward(self, means3D, means2D, opacities, colors_precomp, scales, rotations, shard=None):
        """grad_sync "shard" (round 6): `shard` = this rank's own rows of the six tensors as leaves (shard_rows(P, world, rank)); the
        replicated arguments only feed the rendering and receive no gradient."""
        if self.grad_sync == "shard":
            return _shard_apply(self, False, means3D, opacities, colors_precomp, scales, rotations, shard)
        return _ShellRasterize.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self)


# ======================================================================================================================
# Column wedges: the pixels of a range image are independent, so rank g can own the pixel COLUMNS [e_g, e_{g+1}) (whole
# 16-pixel tile columns) and bin every Gaussian that can reach them.  Its lists are then the complete single-GPU lists of its
# tiles: same instances, same order, same early-out -- the image columns it renders equal a single GPU's up to the GROUPING of the
# per-segment partial sums (a rank derives its segment plan from its own instance total; whenever the plan coincides with the
# single-GPU one -- every configuration the tests and the bench run -- they are bit-identical, which the tests assert), with no
# transmittance exchange and no second pass.  What crosses xGMI:
#   forward   all_gather of the rank's columns of the four image planes (4 H W / N floats per rank; the gradient split sizes ride
#             along) + an all-reduce(max) of the scattered radii, overlapped with the rendering;
#   backward  purely local (pixels outside the wedge have empty lists), then the gradient rows go to their index-chunk owners
#             in ONE variable-split all-to-all as with the shells -- but a Gaussian straddling a boundary has partial rows on
#             both sides, so the owner ADDS rows of equal index (lidargs_wedge_unpack_grad_rows_add); K9/K10 are linear in the
#             per-pixel sums, so adding the ranks' finished rows equals finishing the added sums.
# Cost model against the range shells (lidargs_dist.ShellRasterizer): every stage, the per-pixel ones included, shrinks with N
# (a shell renders the whole image), nothing is walked twice, two collectives fewer sit on the critical path; the price is the
# boundary Gaussians, preprocessed on two ranks (15-25 % at 64 x 2650 over 8 ranks).
# ======================================================================================================================
def wedge_edges(means3D, viewmatrix, W, world, scales=None, shares=None):
    """world + 1 ascending pixel columns, 0 first and W last, interior ones multiples of 16: quantiles of the per-tile-column cost
    (a Gaussian counts once at its projected column, weighted like shell_edges' instance estimate when `scales` is given).
    Load balancing only: any ascending multiples of 16 give the same image and gradients."""
    import math
    V = viewmatrix.reshape(4, 4).to(means3D.dtype)
    p = means3D.detach() @ V[:3, :3] + V[3, :3]
    r = torch.linalg.vector_norm(p, dim=1).clamp(min=1e-3)
    tiles = (W + 15) // 16
    pc = (math.pi - torch.atan2(p[:, 1], p[:, 0])) / (2 * math.pi / W)
    b = (pc / 16.0).long().clamp_(0, tiles - 1)
    if scales is not None:
        a = 3.0 * scales.detach().abs().max(dim=1).values / r
        w = 1.35 + (0.5 * (2.0 * a / (16 * 2 * math.pi / W) + 1.0) * (2.0 * a / 0.022 + 1.0)).clamp(max=15.0)
    else:
        w = torch.ones_like(r)
    cum = torch.cumsum(torch.bincount(b, weights=w.double(), minlength=tiles), 0)
    total = cum[-1].clamp(min=1e-30)
    if shares is None:
        targets = torch.arange(1, world, device=r.device, dtype=cum.dtype) * (total / world)
    else:
        sh = torch.as_tensor(shares, dtype=cum.dtype, device=r.device).clamp(min=1e-6)
        targets = torch.cumsum(sh / sh.sum(), 0)[:-1] * total
    cuts = torch.searchsorted(cum, targets).tolist()
    edges, prev = [0], 0
    for k, i in enumerate(cuts):
        t = min(max(int(i) + 1, prev + 1), tiles - (world - 1 - k))       # strictly ascending, room left for the ranks behind
        edges.append(t * 16); prev = t
    edges.append(W)
    if world > tiles:
        raise ValueError(f"{world} column wedges need at least {world} tile columns; the image has {tiles}")
    return edges


def _wedge_forward(module, means3D, colors, opacities, scales, rotations):
    """Returns ((color, depth, occ, radii), saved-for-backward)."""
    rs, comm, be = module.raster_settings, module.comm, module.backend
    H, W = int(rs.image_height), int(rs.image_width)
    dev = means3D.device
    P = int(means3D.shape[0])
    f32 = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().to(torch.float32).contiguous()
    inp = dict(means3D=f32(means3D), colors=f32(colors), opacities=f32(opacities), scales=f32(scales), rotations=f32(rotations),
               viewmatrix=f32(rs.viewmatrix), beams=f32(rs.beam_inclinations), H=H, W=W, scale_modifier=float(rs.scale_modifier),
               far=int(rs.lidar_far), near=int(rs.lidar_near), bg=rs.bg.to(torch.float32).to(dev).contiguous())
    edges = module.edges
    if edges is None:
        e = torch.tensor(wedge_edges(inp["means3D"], inp["viewmatrix"], W, comm.world, scales=inp["scales"]), dtype=torch.int32, device=dev)
        edges = [int(x) for x in comm.broadcast(e, 0).tolist()]          # every rank must cut at the same columns
    c0, c1 = int(edges[comm.rank]), int(edges[comm.rank + 1])
    wmax = max(int(edges[g + 1]) - int(edges[g]) for g in range(comm.world))

    exchange = comm.world > 1 and module.grad_sync in ("reduce_scatter", "shard")
    fused = getattr(be, "fused_chunk_counts", False)
    enqueue_only = module.enqueue_only and fused
    plan = module.plan.next() if enqueue_only else None
    block = torch.empty(4 * H * wmax + (comm.world if exchange else 0), dtype=torch.float32, device=dev)
    chunks = None
    if exchange:
        rows = _chunk_rows(P, comm.world)
        assert rows < (1 << 24)
        chunks = (rows, comm.world, block[4 * H * wmax:])                          # the all-to-all's split sizes ride on the image gather
    if fused:
        idx, sel = be.select_wedge(inp, c0, c1, plan, chunks=chunks)               # [M], M-row inputs
        st, planes = be.forward_wedge(sel, c0, c1, plan)
    else:
        idx, sel = be.select_wedge(inp, c0, c1)
        if exchange:
            be.chunk_counts(idx, *chunks)
        st, planes = be.forward_wedge(sel, c0, c1)
    radii = be.scatter_radii(idx, st["radii"], P)
    wait_radii = comm.all_reduce_max_async(radii) if comm.world > 1 else (lambda: None)   # a boundary Gaussian reports the same radius twice
    be.pack_columns(planes, H, W, c0, c1, wmax, block)
    blocks = comm.all_gather(block)                                                # [G, 4 H wmax (+ G)]
    color, depth, occ = be.unpack_columns(blocks, edges, H, W, wmax)
    saved = dict(st=st, idx=idx, P=P)
    if exchange:
        saved.update(counts=_HostCounts(blocks[:, 4 * H * wmax:]))                 # [src, dst], read in the backward
    if enqueue_only:
        if plan is None:
            module.plan.learn(int(idx.shape[0]), st["R"])
        else:
            module.plan.submitted()
    wait_radii()
    return (color, depth, occ, radii), saved


def _wedge_backward(module, saved, g_color, g_depth, g_occ):
    st, idx, P = saved["st"], saved["idx"], saved["P"]
    comm, be = module.comm, module.backend
    inp = st["inp"]
    H, W = inp["H"], inp["W"]
    g = be.backward_plain(st, (g_color.reshape(2, H * W), g_depth.reshape(H * W), g_occ.reshape(H * W)))
    sync = module.grad_sync if comm.world > 1 else "none"
    if module.grad_sync == "shard":
        return _shard_grads(module, saved, g, idx, P, add=True)
    if sync == "reduce_scatter":
        dense = be.unpack_rows_add(_exchange_rows(module, saved, g, idx, P), P)
    else:
        packed = be.pack_rows(g, idx)
        module.plan.check()                        # every sync mode (see _shell_backward)
        dense = be.unpack_rows_add(packed, P)
        if sync == "all_reduce":
            dense = comm.all_reduce(dense)
    o, out = 0, {}
    for k, w in GRAD_WIDTHS:
        out[k] = dense[o * P:(o + w) * P].view(P, w)
        o += w
    return out


class _WedgeRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, colors, opacities, scales, rotations, module):
        outs, saved = wedge_forward(module, means3D, colors, opacities, scales, rotations)
        ctx.module, ctx.saved = module, saved
        ctx.mark_non_differentiable(outs[3])
        return outs

    @staticmethod
    def backward(ctx, g_color, g_depth, g_occ, _g_radii):
        g = wedge_backward(ctx.module, ctx.saved, g_color, g_depth, g_occ)
        return g["means3D"], g["means2D"], g["colors"], g["opacities"], g["scales"], g["rotations"], None


class WedgeRasterizer(nn.Module):
    """Column-wedge sharded counterpart of GaussianRasterizer.forward (colors_precomp + scales/rotations path).  Inputs are
    REPLICATED on every rank; outputs are identical on every rank and, for the image, equal to the single-GPU forward's up to the
    grouping of the per-segment partial sums (bit-identical whenever the rank's segment plan is the single-GPU one);
    gradients follow `grad_sync` ("reduce_scatter": rank r ends with rows [r*P/N, (r+1)*P/N); "all_reduce"; "none")."""

    def __init__(self, raster_settings, comm=None, backend=None, grad_sync="reduce_scatter", edges=None):
        super().__init__()
        assert grad_sync in ("reduce_scatter", "all_reduce", "none", "shard")
        self.raster_settings = raster_settings
        self.comm = comm if comm is not None else SingleComm()
        self.backend = backend if backend is not None else HipShellBackend()
        self.grad_sync = grad_sync
        self.edges = edges
        self.enqueue_only = os.environ.get("LIDARGS_ENQUEUE_ONLY", "0") == "1"     # see _RankPlan; off by default
        self.ship_live = _ship_live_default()      # (see ShellRasterizer)
        self.plan = _RankPlan()

    def forward(self, means3D, means2D, opacities, colors_precomp, scales, rotations, shard=None):
        if self.grad_sync == "shard":                # (see ShellRasterizer.forward)
            return _shard_apply(self, True, means3D, opacities, colors_precomp, scales, rotations, shard)
        return _WedgeRasterize.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self)
