"""CPU stand-in for lidargs_dist.HipShellBackend built on the ORACLE (test infrastructure only).

It lets the world_size-2 gloo tests drive the product's collective / compositing logic
(lidargs_dist._ShellRasterize) without a GPU: same protocol, numpy oracle underneath."""
import numpy as np
import torch

from oracle import lgo


class OracleShellBackend:
    def forward(self, inp, lo, hi):
        n = lambda k: inp[k].detach().cpu().numpy()
        f = lgo.forward(n("means3D"), n("colors"), n("opacities"), n("scales"), n("rotations"), n("viewmatrix"), n("beams"),
                        inp["W"], inp["H"], bg=np.zeros(2, np.float32), scale_modifier=inp["scale_modifier"], far=inp["far"], near=inp["near"],
                        shell=(lo, hi), t_only=True)
        st = dict(inp=inp, P=int(inp["means3D"].shape[0]), fwd=f, radii=torch.from_numpy(f.radii.copy()), R=f.num_rendered)
        return st, torch.from_numpy(f.T_pass.copy())

    def render(self, st, T_in):
        f = st["fwd"]
        if st["P"] == 0:
            N = T_in.numel()
            return torch.zeros(3, N), T_in.clone(), T_in.clone()
        lgo.render_shell(f, T_in=T_in.numpy(), t_only=False, bg=None)
        N = T_in.numel()
        part = np.concatenate([f.color.reshape(2, N), f.depth.reshape(1, N)], 0)
        return torch.from_numpy(part.copy()), torch.from_numpy(f.array("final_T").copy()), torch.from_numpy(f.T_pass.copy())

    def backward(self, st, behind, T_final, grads):
        f = st["fwd"]
        gc, gd, go = (g.detach().cpu().numpy() for g in grads)
        g = lgo.backward(f, gc, gd, go, behind=behind.numpy(), T_final_global=T_final.numpy(), bg=st["inp"]["bg"].numpy())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return dict(means3D=t(g["dL_dmeans3D"]), means2D=t(g["dL_dmeans2D"]), colors=t(g["dL_dcolors"]), opacities=t(g["dL_dopacity"]),
                    scales=t(g["dL_dscales"]), rotations=t(g["dL_drotations"]))
