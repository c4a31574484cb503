"""lidargs_dist -- one scene, N GPUs of one node: Gaussians sharded by RANGE SHELL.

The reference is single-GPU; this is new design, shaped by the one property of the path that
matters: front-to-back compositing is order-dependent and every tile's order is by range, so if
rank g owns the Gaussians with range in [e_g, e_{g+1}) each pixel's sorted list is the
concatenation of the ranks' lists.  Per-Gaussian work and gradients are then LOCAL to one rank;
only per-pixel planes cross xGMI:

  forward   1. every rank culls the (replicated) Gaussians to its shell, bins them, and walks its
               lists once for transmittance only                       -> T_pass_g   [N]
            2. all_gather(T_pass)  (N floats/rank; 0.68 MB at 64x2650) -> T_in_g = prod_{h<g} T_pass_h
            3. every rank composites its shell from T_in_g (the reference's global T < 1e-4 early-out
               is applied to the GLOBAL transmittance, so results match the single-GPU walk)
            4. all_gather([C0, C1, D, T_end, T_pass2]_g)  (5 planes/rank) -> image = sum_g partials,
               T_final = T_end of the shell where the walk stopped; also gives each rank what lies
               BEHIND it, which its backward needs
  backward  5. purely local back-to-front pass per shell, seeded with the behind-sums
            6. per-Gaussian gradients have disjoint support across ranks (a Gaussian is in exactly
               one shell): `grad_sync` = "reduce_scatter" (RCCL reduce-scatter of the packed [P,17]
               gradient rows, each rank keeps rows [r*P/N, (r+1)*P/N)), "all_reduce", or "none".

The per-rank compute is behind a small backend protocol so the collective/compositing logic above can
be exercised on CPU with gloo (tests inject a CPU backend); the product backend is HipShellBackend
(C ABI: lidargs_forward_shell / lidargs_render_shell / lidargs_backward_shell).
"""
import ctypes as C

import torch
import torch.nn as nn

GRAD_WIDTHS = (("means3D", 3), ("means2D", 4), ("colors", 2), ("opacities", 1), ("scales", 3), ("rotations", 4))
GRAD_COLS = sum(w for _, w in GRAD_WIDTHS)  # 17 floats = 68 B per Gaussian


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

    def reduce_scatter_rows(self, t):
        return t.clone()


def shell_edges(means3D, viewmatrix, world, near, far, bins=2048):
    """Range-shell boundaries that balance the Gaussian count: world+1 ascending floats, first = -inf,
    last = +inf, interior edges = quantiles of |p_view| from a histogram over (near, far)."""
    V = viewmatrix.reshape(4, 4).to(means3D.dtype)
    p = means3D.detach() @ V[:3, :3] + V[3, :3]
    r = torch.linalg.vector_norm(p, dim=1)
    hist = torch.histc(r, bins=bins, min=float(near), max=float(far))
    cum = torch.cumsum(hist, 0)
    total = cum[-1].clamp(min=1)
    width = (float(far) - float(near)) / bins
    edges = [float("-inf")]
    targets = torch.arange(1, world, device=r.device, dtype=cum.dtype) * (total / world)
    idx = torch.searchsorted(cum, targets)
    for i in idx.tolist():
        edges.append(float(near) + (int(i) + 1) * width)
    edges.append(float("inf"))
    return torch.tensor(edges, dtype=torch.float32, device=means3D.device)


class HipShellBackend:
    """Per-rank compute on a HIP device through the C ABI (include/lidargs_rasterizer.h)."""

    def __init__(self):
        from diff_lidargs_rasterization import _C
        self._C = _C
        self.lib = _C._lib

    def forward(self, inp, lo, hi):
        _C, lib = self._C, self.lib
        m3 = inp["means3D"]
        _C._require_device(m3, "means3D")
        dev, P, H, W = m3.device, int(m3.shape[0]), inp["H"], inp["W"]
        st = dict(inp=inp, P=P, geom=_C._Scratch(dev), binning=_C._Scratch(dev), img=_C._Scratch(dev))
        st["radii"] = torch.zeros(P, dtype=torch.int32, device=dev)
        st["radii_xy"] = torch.zeros(2 * P, dtype=torch.int32, device=dev)
        T_pass = torch.ones(H * W, dtype=torch.float32, device=dev)
        dummy = torch.zeros(4 * H * W, dtype=torch.float32, device=dev)
        n = 0
        if P:
            p = _C._ptr
            with torch.cuda.device(dev):
                n = lib.lidargs_forward_shell(
                    _C._alloc_cb, st["geom"].user, _C._alloc_cb, st["binning"].user, _C._alloc_cb, st["img"].user, C.c_int(P), None,
                    C.c_int(W), C.c_int(H),
                    p(m3), p(inp["colors"]), p(inp["opacities"]), p(inp["scales"]), C.c_float(inp["scale_modifier"]),
                    p(inp["rotations"]), None, p(inp["viewmatrix"]), p(inp["beams"]), C.c_int(inp["far"]), C.c_int(inp["near"]),
                    C.c_float(lo), C.c_float(hi), None, C.c_int(1), p(dummy), p(dummy[2 * H * W:]), p(dummy[3 * H * W:]), p(T_pass),
                    p(st["radii"]), p(st["radii_xy"]), C.c_int(0), _C._stream(dev))
            if n < 0:
                _C._raise(n, "lidargs_forward_shell")
        for k in ("geom", "binning", "img"):       # keep only the tensors: nothing holds the registry entries alive
            st[k] = st[k].take()
        st["R"] = n
        return st, T_pass

    def render(self, st, T_in):
        _C, lib = self._C, self.lib
        inp = st["inp"]
        dev, H, W = T_in.device, inp["H"], inp["W"]
        N = H * W
        part = torch.zeros(4 * N, dtype=torch.float32, device=dev)      # colour0, colour1, depth, (occ scratch)
        T_pass2 = T_in.clone()
        T_end = T_in.clone()
        if st["P"]:
            p = _C._ptr
            with torch.cuda.device(dev):
                rc = lib.lidargs_render_shell(C.c_int(st["P"]), C.c_int(st["R"]), None, C.c_int(W), C.c_int(H), p(st["geom"]),
                                              p(st["binning"]), p(st["img"]), p(T_in.contiguous()), C.c_int(0), p(part),
                                              p(part[2 * N:]), p(part[3 * N:]), p(T_pass2), p(T_end), C.c_int(0), _C._stream(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_render_shell")
        return part[:3 * N].view(3, N), T_end, T_pass2

    def backward(self, st, behind, T_final, grads):
        _C, lib = self._C, self.lib
        inp = st["inp"]
        P, H, W = st["P"], inp["H"], inp["W"]
        dev = behind.device
        widths = (3, 4, 2, 1, 4, 1, 6, 3, 4, 3, 3, 3)
        slab = torch.zeros(P * sum(widths), dtype=torch.float32, device=dev)
        parts, o = [], 0
        for w in widths:
            parts.append(slab[o:o + P * w].view(P, w)); o += P * w
        (g_m3, g_m2, g_col, g_dep, g_con, g_op, g_cov, g_sc, g_rot, g_sph, g_u1, g_u2) = parts
        if P:
            p = _C._ptr
            gc, gd, go = (g.contiguous() for g in grads)
            with torch.cuda.device(dev):
                rc = lib.lidargs_backward_shell(
                    C.c_int(P), C.c_int(st["R"]), p(inp["bg"]), C.c_int(W), C.c_int(H), p(inp["means3D"]), p(inp["colors"]), p(inp["scales"]),
                    C.c_float(inp["scale_modifier"]), p(inp["rotations"]), None, p(inp["viewmatrix"]), p(inp["beams"]), p(st["radii"]),
                    p(st["geom"]), p(st["binning"]), p(st["img"]), p(behind.contiguous()), p(T_final.contiguous()),
                    p(gc), p(gd), p(go), p(g_m2), p(g_con), p(g_op), p(g_col), p(g_dep), p(g_m3), p(g_sph), p(g_u1), p(g_u2), p(g_cov),
                    p(g_sc), p(g_rot), C.c_int(0), _C._stream(dev))
            if rc < 0:
                _C._raise(rc, "lidargs_backward_shell")
        return dict(means3D=g_m3, means2D=g_m2, colors=g_col, opacities=g_op, scales=g_sc, rotations=g_rot)


def shell_forward(module, means3D, colors, opacities, scales, rotations):
    """Steps 1-4 of the module docstring.  Returns ((color, depth, occ, radii), saved-for-backward)."""
    rs, comm, be = module.raster_settings, module.comm, module.backend
    H, W = int(rs.image_height), int(rs.image_width)
    N = H * W
    dev = means3D.device
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    inp = dict(means3D=f32(means3D), colors=f32(colors), opacities=f32(opacities), scales=f32(scales), rotations=f32(rotations),
               viewmatrix=f32(rs.viewmatrix), beams=f32(rs.beam_inclinations), H=H, W=W, scale_modifier=float(rs.scale_modifier),
               far=int(rs.lidar_far), near=int(rs.lidar_near), bg=rs.bg.to(torch.float32).to(dev).contiguous())
    edges = module.edges
    if edges is None:
        edges = shell_edges(inp["means3D"], inp["viewmatrix"], comm.world, rs.lidar_near, rs.lidar_far)
        edges = comm.broadcast(edges, 0)       # every rank must cut at the same ranges
    lo, hi = float(edges[comm.rank]), float(edges[comm.rank + 1])

    st, T_pass = be.forward(inp, lo, hi)                                          # 1
    allT = comm.all_gather(T_pass)                                                # 2   [G, N]
    T_in = torch.ones(N, dtype=torch.float32, device=dev)
    if comm.rank > 0:
        T_in = torch.prod(allT[:comm.rank], dim=0)
    part, T_end, T_pass2 = be.render(st, T_in)                                    # 3
    planes = comm.all_gather(torch.cat([part, T_end.view(1, N), T_pass2.view(1, N)], 0))   # 4   [G, 5, N]
    img = planes[:, :3].sum(0)
    # the walk stopped in the first shell whose hand-over value fell below the reference's 1e-4 threshold
    stopped = planes[:, 4] < 1e-4                                                 # [G, N]
    first = torch.where(stopped.any(0), stopped.float().argmax(0), torch.full((N,), comm.world - 1, device=dev))
    T_final = planes[:, 3].gather(0, first.view(1, N)).view(N)
    bg = inp["bg"]
    color = torch.stack([img[0] + T_final * bg[0], img[1] + T_final * bg[1]], 0).view(2, H, W)
    depth = img[2].view(1, H, W)
    occ = (1.0 - T_final).view(1, H, W)
    behind = planes[comm.rank + 1:, :3].sum(0) if comm.rank + 1 < comm.world else torch.zeros(3, N, dtype=torch.float32, device=dev)
    radii = st["radii"]
    if comm.world > 1:                          # radii of the other shells' Gaussians
        radii = comm.all_reduce(radii.clone())
    return (color, depth, occ, radii), (st, behind.contiguous(), T_final)


def shell_backward(module, saved, g_color, g_depth, g_occ):
    """Steps 5-6.  Returns {means3D, means2D, colors, opacities, scales, rotations} gradients."""
    st, behind, T_final = saved
    comm, be = module.comm, module.backend
    inp = st["inp"]
    H, W, P = inp["H"], inp["W"], st["P"]
    # d(color)/d(T_final) for the background is inside the blend: (-T_final/(1-alpha)) * bg.g  (R3/cr/backward.cu:727)
    g = be.backward(st, behind, T_final, (g_color.reshape(2, H * W), g_depth.reshape(H * W), g_occ.reshape(H * W)))
    if comm.world > 1 and module.grad_sync != "none":
        packed = torch.cat([g[k] for k, _ in GRAD_WIDTHS], dim=1)                 # [P, 17]
        if module.grad_sync == "all_reduce":
            packed = comm.all_reduce(packed)
        else:
            rows = (P + comm.world - 1) // comm.world
            pad = rows * comm.world - P
            if pad:
                packed = torch.cat([packed, packed.new_zeros(pad, GRAD_COLS)], 0)
            mine = comm.reduce_scatter_rows(packed)                               # 6
            packed = packed.new_zeros(rows * comm.world, GRAD_COLS)
            packed[comm.rank * rows:(comm.rank + 1) * rows] = mine
            packed = packed[:P]
        o, out = 0, {}
        for k, w in GRAD_WIDTHS:
            out[k] = packed[:, o:o + w].contiguous(); o += w
        g = out
    return g


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


class ShellRasterizer(nn.Module):
    """Range-shell sharded counterpart of GaussianRasterizer.forward (colors_precomp + scales/rotations path,
    the one gaussian_renderer.render() uses).  Inputs are REPLICATED on every rank; outputs are identical
    on every rank; gradients follow `grad_sync`."""

    def __init__(self, raster_settings, comm=None, backend=None, grad_sync="reduce_scatter", edges=None):
        super().__init__()
        assert grad_sync in ("reduce_scatter", "all_reduce", "none")
        self.raster_settings = raster_settings
        self.comm = comm if comm is not None else SingleComm()
        self.backend = backend if backend is not None else HipShellBackend()
        self.grad_sync = grad_sync
        self.edges = edges

    def forward(self, means3D, means2D, opacities, colors_precomp, scales, rotations):
        return _ShellRasterize.apply(means3D, means2D, colors_precomp, opacities, scales, rotations, self)
