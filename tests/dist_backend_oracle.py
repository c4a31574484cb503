"""CPU stand-in for lidargs_dist.HipShellBackend built on the ORACLE (test infrastructure only).

It lets the world_size-2 gloo tests drive the product's collective / compositing logic
(lidargs_dist._ShellRasterize) without a GPU: same protocol, numpy oracle underneath."""
import numpy as np
import torch

from oracle import lgo


ROW_KEYS = ("means3D", "colors", "opacities", "scales", "rotations")


class OracleShellBackend:
    def select(self, inp, lo, hi):
        """Same float arithmetic as the oracle's own shell test (K1 view transform, left-to-right fp32, then sqrtf)."""
        m = inp["means3D"].detach().cpu().numpy().astype(np.float32)
        v = inp["viewmatrix"].detach().cpu().numpy().astype(np.float32).reshape(16)
        x, y, z = m[:, 0], m[:, 1], m[:, 2]
        px = ((v[0] * x + v[4] * y) + v[8] * z) + v[12]
        py = ((v[1] * x + v[5] * y) + v[9] * z) + v[13]
        pz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14]
        dist = np.sqrt((px * px + py * py) + pz * pz, dtype=np.float32)
        keep = (dist >= np.float32(lo)) & (dist < np.float32(hi))
        idx = torch.from_numpy(np.nonzero(keep)[0].astype(np.int32))
        sel = dict(inp)
        for k in ROW_KEYS:
            sel[k] = inp[k][idx.long()]
        return idx, sel

    # step 6 helpers: same protocol as HipShellBackend, in framework ops
    def scatter_radii(self, idx, radii_shell, P):
        out = torch.zeros(P, dtype=torch.int32)
        out[idx.long()] = radii_shell
        return out

    def chunk_counts(self, idx, chunk_rows, world, out):
        bounds = torch.arange(0, world + 1, dtype=idx.dtype) * chunk_rows
        cuts = torch.searchsorted(idx, bounds)
        out.copy_((cuts[1:] - cuts[:-1]).to(torch.float32))

    def pack_rows(self, g, idx):
        cols = [g[k] for k in ("means3D", "means2D", "colors", "opacities", "scales", "rotations")]
        return torch.cat(cols + [idx.view(torch.float32).view(-1, 1)], dim=1)

    def _live(self, g, idx, P, chunk_rows, world):
        rows = self.pack_rows(g, idx)
        ii = idx.long()
        live = (rows[:, :17] != 0).any(1) & (ii >= 0) & (ii < P)
        dest = torch.clamp(ii // int(chunk_rows), max=world - 1)
        return rows[live], dest[live]

    def count_rows_live(self, g, idx, P, chunk_rows, world):
        """Rows with a non-zero gradient per destination chunk, as lidargs_shell_pack_grad_rows_live_count (int32 [2 * world], the first half counts)."""
        _, dest = self._live(g, idx, P, chunk_rows, world)
        return torch.cat([torch.bincount(dest, minlength=world).to(torch.int32), torch.zeros(world, dtype=torch.int32)])

    def pack_rows_live(self, g, idx, P, chunk_rows, world, cnt, n):
        rows, dest = self._live(g, idx, P, chunk_rows, world)
        assert rows.shape[0] == n
        return rows[torch.argsort(dest, stable=True)]

    def unpack_rows(self, rows, P, blocked=False):
        dense = torch.zeros((P, 17), dtype=torch.float32)
        if rows.shape[0]:
            dense.index_copy_(0, rows[:, 17].contiguous().view(torch.int32).long(), rows[:, :17])
        if not blocked:
            return dense
        o, parts = 0, []
        for w in (3, 4, 2, 1, 3, 4):                      # six contiguous blocks [P, w]
            parts.append(dense[:, o:o + w].contiguous().view(-1)); o += w
        return torch.cat(parts)

    def unpack_rows_chunk(self, rows, base, chunk_rows, add=False):
        """[n, 18] rows -> the [17 * chunk_rows] block of the rows [base, base + chunk_rows) (six contiguous gradient blocks), as
        lidargs_shell_unpack_grad_rows_chunk; add: rows of equal index are added."""
        n = int(chunk_rows)
        dense = torch.zeros(n, 17)
        if rows.shape[0]:
            g = rows[:, 17].contiguous().view(torch.int32).long() - int(base)
            ok = (g >= 0) & (g < n)
            if add:
                dense.index_add_(0, g[ok], rows[ok, :17])
            else:
                dense[g[ok]] = rows[ok, :17]
        out, o = [], 0
        for w in (3, 4, 2, 1, 3, 4):
            out.append(dense[:, o:o + w].contiguous().view(-1)); o += w
        return torch.cat(out)

    def transmittance(self, allT, rank):
        return torch.prod(allT[:rank], dim=0) if rank > 0 else torch.ones_like(allT[0])

    def compose(self, planes, rank, bg, H, W):
        G, N = planes.shape[0], H * W
        img = planes[:, :3].sum(0)
        stopped = planes[:, 4] < 1e-4
        first = torch.where(stopped.any(0), stopped.float().argmax(0), torch.full((N,), G - 1))
        T_final = planes[:, 3].gather(0, first.view(1, N)).view(N)
        color = torch.stack([img[0] + T_final * bg[0], img[1] + T_final * bg[1]], 0).view(2, H, W)
        behind = planes[rank + 1:, :3].sum(0) if rank + 1 < G else torch.zeros(3, N)
        return color, img[2].view(1, H, W), (1.0 - T_final).view(1, H, W), T_final, behind.contiguous()

    def forward(self, inp, lo, hi):
        if int(inp["means3D"].shape[0]) == 0:        # empty shell: nothing in the way
            return dict(inp=inp, P=0, fwd=None, radii=torch.zeros(0, dtype=torch.int32), R=0), torch.ones(inp["H"] * inp["W"])
        n = lambda k: inp[k].detach().cpu().numpy()
        f = lgo.forward(n("means3D"), n("colors"), n("opacities"), n("scales"), n("rotations"), n("viewmatrix"), n("beams"),
                        inp["W"], inp["H"], bg=np.zeros(2, np.float32), scale_modifier=inp["scale_modifier"], far=inp["far"], near=inp["near"],
                        shell=(lo, hi), t_only=True)
        st = dict(inp=inp, P=int(inp["means3D"].shape[0]), fwd=f, radii=torch.from_numpy(f.radii.copy()), R=f.num_rendered)
        return st, torch.from_numpy(f.T_pass.copy())

    def render(self, st, T_in):
        f = st["fwd"]
        N = T_in.numel()
        if st["P"] == 0:
            return torch.cat([torch.zeros(3, N), T_in.view(1, N), T_in.view(1, N)], 0)
        lgo.render_shell(f, T_in=T_in.numpy(), t_only=False, bg=None)
        planes = np.concatenate([f.color.reshape(2, N), f.depth.reshape(1, N), f.array("final_T").reshape(1, N), f.T_pass.reshape(1, N)], 0)
        return torch.from_numpy(planes.astype(np.float32).copy())

    def backward(self, st, behind, T_final, grads):
        if st["P"] == 0:
            return {k: torch.zeros(0, w) for k, w in (("means3D", 3), ("means2D", 4), ("colors", 2), ("opacities", 1), ("scales", 3), ("rotations", 4))}
        f = st["fwd"]
        gc, gd, go = (g.detach().cpu().numpy() for g in grads)
        g = lgo.backward(f, gc, gd, go, behind=behind.numpy(), T_final_global=T_final.numpy(), bg=st["inp"]["bg"].numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return dict(means3D=t(g["dL_dmeans3D"]), means2D=t(g["dL_dmeans2D"]), colors=t(g["dL_dcolors"]), opacities=t(g["dL_dopacity"]),
                    scales=t(g["dL_dscales"]), rotations=t(g["dL_drotations"]))


class OracleWedgeBackend(OracleShellBackend):
    """CPU stand-in for the column-wedge half of HipShellBackend: the selection restates k_wedge_flags' bound (so a bound that is
    too tight shows up as a wrong image here), the render is the oracle on the selected rows with everything outside the rank's
    pixel columns ignored: cropped in the forward, upstream gradients zeroed in the backward."""

    def select_wedge(self, inp, c0, c1):
        m = inp["means3D"].detach().cpu().numpy().astype(np.float32)
        v = inp["viewmatrix"].detach().cpu().numpy().astype(np.float32).reshape(16)
        W = inp["W"]
        x, y, z = m[:, 0], m[:, 1], m[:, 2]
        px = ((v[0] * x + v[4] * y) + v[8] * z) + v[12]
        py = ((v[1] * x + v[5] * y) + v[9] * z) + v[13]
        pz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14]
        d2 = (px * px + py * py) + pz * pz
        smax = np.float32(inp["scale_modifier"]) * np.abs(inp["scales"].detach().cpu().numpy()).max(1)
        q = inp["rotations"].detach().cpu().numpy().astype(np.float32)
        nq = np.maximum(1.0, (q * q).sum(1))
        A = (smax * smax * nq * nq * 1.0001 + 0.01) / np.maximum(d2, 1e-12)
        step = np.float32(2 * np.float32(np.pi) / W)
        rx = 3.0 * np.sqrt(2.0 * A + 3.2e-5) / np.tan(step) * 1.001 + 1.0
        pc = (np.float32(np.pi) - np.arctan2(py, px)) / step
        reach = rx + 18.0
        keep = (pc + reach >= c0) & (pc - reach < c1) & (d2 > 0)
        idx = torch.from_numpy(np.nonzero(keep)[0].astype(np.int32))
        sel = dict(inp)
        for k in ROW_KEYS:
            sel[k] = inp[k][idx.long()]
        return idx, sel

    def forward_wedge(self, inp, c0, c1):
        H, W = inp["H"], inp["W"]
        N = H * W
        P = int(inp["means3D"].shape[0])
        bg = inp["bg"].numpy()
        if P == 0:
            planes = torch.zeros(4, N)
            planes[0] = float(bg[0]); planes[1] = float(bg[1])
            return dict(inp=inp, P=0, fwd=None, radii=torch.zeros(0, dtype=torch.int32), R=0, cols=(c0, c1)), planes.view(-1)
        n = lambda k: inp[k].detach().cpu().numpy()
        f = lgo.forward(n("means3D"), n("colors"), n("opacities"), n("scales"), n("rotations"), n("viewmatrix"), n("beams"), W, H, bg=bg,
                        scale_modifier=inp["scale_modifier"], far=inp["far"], near=inp["near"])
        planes = np.concatenate([f.color.reshape(2, N), f.depth.reshape(1, N), f.occ.reshape(1, N)], 0).astype(np.float32)
        st = dict(inp=inp, P=P, fwd=f, radii=torch.from_numpy(f.radii.copy()), R=f.num_rendered, cols=(c0, c1))
        return st, torch.from_numpy(planes.copy()).view(-1)

    def backward_plain(self, st, grads):
        if st["P"] == 0:
            return {k: torch.zeros(0, w) for k, w in (("means3D", 3), ("means2D", 4), ("colors", 2), ("opacities", 1), ("scales", 3), ("rotations", 4))}
        H, W = st["inp"]["H"], st["inp"]["W"]
        c0, c1 = st["cols"]
        gc, gd, go = (g.detach().cpu().numpy().copy() for g in grads)
        mask = np.zeros((H, W), np.float32); mask[:, c0:c1] = 1.0         # only the rank's own pixels exist for it
        gc = gc.reshape(2, H, W) * mask; gd = gd.reshape(H, W) * mask; go = go.reshape(H, W) * mask
        g = lgo.backward(st["fwd"], gc, gd, go)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return dict(means3D=t(g["dL_dmeans3D"]), means2D=t(g["dL_dmeans2D"]), colors=t(g["dL_dcolors"]), opacities=t(g["dL_dopacity"]),
                    scales=t(g["dL_dscales"]), rotations=t(g["dL_drotations"]))

    def pack_columns(self, planes, H, W, c0, c1, wmax, out):
        blk = torch.zeros(4, H, wmax)
        blk[:, :, :c1 - c0] = planes.view(4, H, W)[:, :, c0:c1]
        out[:4 * H * wmax] = blk.view(-1)

    def unpack_columns(self, blocks, edges, H, W, wmax):
        full = torch.zeros(4, H, W)
        for g in range(blocks.shape[0]):
            e0, e1 = int(edges[g]), int(edges[g + 1])
            full[:, :, e0:e1] = blocks[g, :4 * H * wmax].view(4, H, wmax)[:, :, :e1 - e0]
        return full[:2].contiguous(), full[2:3].contiguous(), full[3:4].contiguous()

    def unpack_rows_add(self, rows, P):
        dense = torch.zeros((P, 17), dtype=torch.float32)
        if rows.shape[0]:
            dense.index_add_(0, rows[:, 17].contiguous().view(torch.int32).long(), rows[:, :17])
        o, parts = 0, []
        for w in (3, 4, 2, 1, 3, 4):
            parts.append(dense[:, o:o + w].contiguous().view(-1)); o += w
        return torch.cat(parts)
