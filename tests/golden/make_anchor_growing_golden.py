"""Generates tests/golden/anchor_growing_golden.npz by EXECUTING the reference's GaussianModel.anchor_growing
(/root/reference/scene/gaussian_model.py:677-775) together with its own cat_tensors_to_optimizer (:567-594) and inverse_sigmoid
(utils/general_utils.py:19-20) on CPU torch.  The methods are taken out of the parsed class by name (the module cannot be imported
here: torch_scatter, simple_knn, plyfile) and bound to a small host class that carries the tensors, a real torch.optim.Adam with the
reference's parameter-group names (:385-391), and the two properties the method reads (get_anchor :255, get_scaling :213-214).

What is NOT the reference's code, stated as such:
  * `scatter_max` -- torch_scatter is absent from this image.  Stand-in (three lines, below): Tensor.scatter_reduce(0, index, src,
    "amax", include_self=False), returning (values, None) like torch_scatter does.  The feature maximum is therefore pinned against
    torch's own reduction, not against torch_scatter's kernel.
  * device placement: `.cuda()` and device="cuda" are re-targeted to the CPU by a TorchFunctionMode (no GPU in this container);
    the same mode records what torch.rand_like returned at each level (:687) so that the fixture holds the random draws.

    python tests/golden/make_anchor_growing_golden.py
"""
import ast
import os
from functools import reduce

import numpy as np
import torch
from torch import nn
from torch.overrides import TorchFunctionMode

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "anchor_growing_golden.npz")


def scatter_max(src, index, dim=0):
    """STAND-IN for torch_scatter.scatter_max (absent here): per-index maximum over dim 0, (values, argmax=None)."""
    n = int(index.max()) + 1 if index.numel() else 0
    return torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype).scatter_reduce(dim, index, src, "amax", include_self=False), None


class CpuAndRecord(TorchFunctionMode):
    def __init__(self):
        super().__init__()
        self.rands = []

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if str(kwargs.get("device", "")).startswith("cuda"):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        out = func(*args, **kwargs)
        if func is torch.rand_like:
            self.rands.append(out.clone())
        return out


def reference_methods():
    tree = ast.parse(open(os.path.join(REF, "scene/gaussian_model.py")).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel")
    names = ("anchor_growing", "cat_tensors_to_optimizer", "adjust_anchor", "prune_anchor", "_prune_anchor_optimizer")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names)
    util = ast.parse(open(os.path.join(REF, "utils/general_utils.py")).read())
    inv = next(n for n in util.body if isinstance(n, ast.FunctionDef) and n.name == "inverse_sigmoid")
    ns = {"torch": torch, "nn": nn, "reduce": reduce, "scatter_max": scatter_max}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[inv] + fns, type_ignores=[])), REF, "exec"), ns)
    return ns


def host_class():
    ns = reference_methods()

    class Host:
        anchor_growing = ns["anchor_growing"]
        cat_tensors_to_optimizer = ns["cat_tensors_to_optimizer"]
        adjust_anchor = ns["adjust_anchor"]                                                # :776-830 (case "adj")
        prune_anchor = ns["prune_anchor"]
        _prune_anchor_optimizer = ns["_prune_anchor_optimizer"]
        get_anchor = property(lambda self: self._anchor)                                   # scene/gaussian_model.py:254-256
        get_scaling = property(lambda self: 1.0 * torch.exp(self._scaling))                # :212-214 (scaling_activation = torch.exp, :39)
    return Host


PARAMS = ("anchor", "offset", "anchor_feat", "opacity", "scaling", "rotation")


def scene(tag, seed):
    """Inputs of a case.  Anchors sit on the voxel grid like the reference's initialisation leaves them (:274: unique(round(p / v)) * v)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g)
    if tag == "a":       # 0.01-m voxels (not a power of two: the quotient convention matters), dense cluster, all three levels grow
        N, k, voxel, thr = 600, 6, 0.01, 0.0005
        cells = torch.unique(torch.round((r(N * 2, 3) - 0.5) * torch.tensor([6.0, 6.0, 1.0]) / voxel), dim=0)
        anchor = (cells[torch.randperm(cells.shape[0], generator=g)[:N]] * voxel).float()
        offset = (r(N, k, 3) - 0.5) * 2.0
        scaling = torch.log(r(N, 6) * 0.3 + 0.02)
        grads = r(N * k) * 0.004
    elif tag == "b":     # 1/16-m voxels, k = 10: every quotient is exact, offsets placed ON half-integer voxel coordinates (round half to even),
                         # many candidates per voxel (the feature maximum) and candidates falling into voxels that already hold an anchor
        N, k, voxel, thr = 300, 10, 0.0625, 0.0002
        cells = torch.unique(torch.randint(-6, 7, (N * 3, 3), generator=g), dim=0)
        anchor = (cells[torch.randperm(cells.shape[0], generator=g)[:N]].float() * voxel * 4)
        N = anchor.shape[0]
        offset = torch.randint(-12, 13, (N, k, 3), generator=g).float() * 0.5               # in units of the scaling below
        scaling = torch.log(torch.full((N, 6), voxel))                                      # exp(log(2^-4)) need not be exact in float32: recorded as computed
        grads = r(N * k) * 0.002
    elif tag == "c":     # level 0 grows nothing (its threshold is out of reach) -> levels 1 and 2 are skipped (:691-693)
        N, k, voxel, thr = 200, 5, 0.02, 0.5
        anchor = torch.round((r(N, 3) - 0.5) * 4 / voxel) * voxel
        offset = (r(N, k, 3) - 0.5)
        scaling = torch.log(r(N, 6) * 0.2 + 0.05)
        grads = r(N * k) * 0.01
    else:
        raise ValueError(tag)
    N = anchor.shape[0]
    feat = torch.randn(N, 32, generator=g)
    feat[::7, ::5] = 0.0
    feat[3::11, 1::4] *= -1
    offset_mask = r(N * k) > 0.2
    # values sitting exactly on the float32 thresholds of levels 0 and 1 (`>=` must take them: the Python double is cast to float32, :683)
    grads[:4] = torch.tensor([thr, thr * 2], dtype=torch.float32).repeat(2)
    offset_mask[:4] = True
    return dict(N=N, k=k, voxel=voxel, thr=thr, anchor=anchor.float(), offset=offset.float(), scaling=scaling.float(), feat=feat.float(), grads=grads.float(),
                offset_mask=offset_mask, seed=seed)


def run_adjust(tag, seed, out):
    """GaussianModel.adjust_anchor as a whole (:776-830): gradient norms from the accumulators, anchor_growing, the statistics' reset and
    padding, the prune masks, prune_anchor with the optimizer surgery -- executed on case `a`'s model with accumulators built to make
    every branch run (offsets over and under the visit threshold, NaN gradients from 0 / 0, anchors to prune)."""
    c = scene("a", seed)
    Host = host_class()
    h = Host()
    N, k = c["N"], c["k"]
    h.n_offsets, h.feat_dim, h.voxel_size = k, 32, c["voxel"]
    h.update_depth, h.update_init_factor, h.update_hierachy_factor = 3, 16, 4
    h._anchor = nn.Parameter(c["anchor"].clone()); h._offset = nn.Parameter(c["offset"].clone()); h._anchor_feat = nn.Parameter(c["feat"].clone())
    h._opacity = nn.Parameter(torch.full((N, 1), 0.25)); h._scaling = nn.Parameter(c["scaling"].clone())
    h._rotation = nn.Parameter(torch.tensor([[1.0, 0, 0, 0]]).repeat(N, 1))
    h.optimizer = torch.optim.Adam([{"params": [getattr(h, "_" + n)], "lr": 1e-3, "name": n} for n in PARAMS], lr=0.0, eps=1e-15)
    for n in PARAMS:
        getattr(h, "_" + n).grad = torch.zeros_like(getattr(h, "_" + n))
    h.optimizer.step()
    g = torch.Generator().manual_seed(seed + 5)
    denom = torch.randint(0, 40, (N * k, 1), generator=g).float()                           # visits; > check_interval * success_threshold = 10 passes :781
    accum = c["grads"].view(-1, 1) * denom                                                  # so that accum / denom = the case's gradient norms (0 / 0 -> NaN -> 0, :779)
    h.offset_gradient_accum, h.offset_denom = accum.clone(), denom.clone()
    h.anchor_demon = torch.randint(0, 40, (N, 1), generator=g).float()
    h.opacity_accum = torch.rand(N, 1, generator=g) * 0.4 * h.anchor_demon * 0.02           # some below min_opacity * anchor_demon (:799)
    ins = dict(offset_gradient_accum=accum, offset_denom=denom, anchor_demon=h.anchor_demon.clone(), opacity_accum=h.opacity_accum.clone())
    torch.manual_seed(c["seed"] + 2000)
    mode = CpuAndRecord()
    with torch.no_grad(), mode:
        h.adjust_anchor(check_interval=100, success_threshold=0.1, grad_threshold=c["thr"], min_opacity=0.005)     # train.py:247 with arguments/__init__.py:150-155
    npy = lambda t: t.detach().numpy().copy()
    out.update({f"{tag}_N": N, f"{tag}_k": k, f"{tag}_voxel_size": np.float64(c["voxel"]), f"{tag}_threshold": np.float64(c["thr"]),
                f"{tag}_in_anchor": npy(c["anchor"]), f"{tag}_in_offset": npy(c["offset"]), f"{tag}_in_scaling": npy(c["scaling"]), f"{tag}_in_anchor_feat": npy(c["feat"]),
                f"{tag}_n_rand": len(mode.rands)})
    for n, t in ins.items():
        out[f"{tag}_in_{n}"] = npy(t)
    for i, rr in enumerate(mode.rands):
        out[f"{tag}_rand{i}"] = npy(rr)
    for n in PARAMS:
        out[f"{tag}_out_{n}"] = npy(getattr(h, "_" + n))
    for n in ("offset_gradient_accum", "offset_denom", "anchor_demon", "opacity_accum", "max_radii2D"):
        out[f"{tag}_out_{n}"] = npy(getattr(h, n))
    print(tag, "anchors", N, "->", h._anchor.shape[0], "rand draws", len(mode.rands))


def run(tag, seed, out):
    c = scene(tag, seed)
    Host = host_class()
    h = Host()
    h.n_offsets, h.feat_dim, h.voxel_size = c["k"], 32, c["voxel"]
    h.update_depth, h.update_init_factor, h.update_hierachy_factor = 3, 16, 4               # arguments/__init__.py:55-57
    N, k = c["N"], c["k"]
    h._anchor = nn.Parameter(c["anchor"].clone())
    h._offset = nn.Parameter(c["offset"].clone())
    h._anchor_feat = nn.Parameter(c["feat"].clone())
    h._opacity = nn.Parameter(torch.full((N, 1), 0.25))
    h._scaling = nn.Parameter(c["scaling"].clone())
    h._rotation = nn.Parameter(torch.tensor([[1.0, 0, 0, 0]]).repeat(N, 1))
    h.optimizer = torch.optim.Adam([{"params": [getattr(h, "_" + n)], "lr": 1e-3, "name": n} for n in PARAMS], lr=0.0, eps=1e-15)   # :385-391
    for n in PARAMS:                                                                        # one step so that exp_avg / exp_avg_sq exist (the branch :575-582)
        getattr(h, "_" + n).grad = torch.zeros_like(getattr(h, "_" + n))
    h.optimizer.step()
    h.anchor_demon = torch.arange(N, dtype=torch.float32).view(N, 1)
    h.opacity_accum = torch.arange(N, dtype=torch.float32).view(N, 1) * 0.5
    torch.manual_seed(c["seed"] + 1000)
    mode = CpuAndRecord()
    with torch.no_grad(), mode:
        h.anchor_growing(c["grads"].clone(), c["thr"], c["offset_mask"].clone())
    npy = lambda t: t.detach().numpy().copy()
    out.update({f"{tag}_N": N, f"{tag}_k": k, f"{tag}_voxel_size": np.float64(c["voxel"]), f"{tag}_threshold": np.float64(c["thr"]),
                f"{tag}_in_anchor": npy(c["anchor"]), f"{tag}_in_offset": npy(c["offset"]), f"{tag}_in_scaling": npy(c["scaling"]), f"{tag}_in_anchor_feat": npy(c["feat"]),
                f"{tag}_grads": npy(c["grads"]), f"{tag}_offset_mask": npy(c["offset_mask"]), f"{tag}_n_rand": len(mode.rands)})
    for i, rr in enumerate(mode.rands):
        out[f"{tag}_rand{i}"] = npy(rr)
    for n in PARAMS:
        out[f"{tag}_out_{n}"] = npy(getattr(h, "_" + n))
        st = h.optimizer.state[getattr(h, "_" + n)]
        assert st["exp_avg"].shape == getattr(h, "_" + n).shape
    out[f"{tag}_out_anchor_demon"] = npy(h.anchor_demon)
    out[f"{tag}_out_opacity_accum"] = npy(h.opacity_accum)
    print(tag, "anchors", N, "->", h._anchor.shape[0], "rand draws", len(mode.rands))


if __name__ == "__main__":
    out = {}
    run("a", 11, out)
    run("b", 12, out)
    run("c", 13, out)
    run_adjust("adj", 14, out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
