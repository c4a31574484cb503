"""Generates tests/golden/training_statis_golden.npz by EXECUTING the reference's GaussianModel.training_statis
(/root/reference/scene/gaussian_model.py:599-622) on CPU torch.  The method is taken out of the parsed class by name (the module
itself cannot be imported here: torch_scatter, simple_knn, plyfile) and called with a plain namespace as `self`.

    python tests/golden/make_training_statis_golden.py
"""
import ast
import os
import types

import numpy as np
import torch

REF = "/root/reference/scene/gaussian_model.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "training_statis_golden.npz")


def reference_method():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "training_statis")
    ns = {"torch": torch}
    exec(compile(ast.fix_missing_locations(ast.Module(body=[fn], type_ignores=[])), REF, "exec"), ns)
    return ns["training_statis"]


def run(tag, N, k, seed, out):
    g = torch.Generator().manual_seed(seed)
    vis = torch.rand(N, generator=g) > 0.3
    n = int(vis.sum())
    opacity = torch.rand(n * k, 1, generator=g) * 2 - 1                      # tanh range
    sel = (opacity > 0).view(-1)
    M = int(sel.sum())
    update_filter = torch.rand(M, generator=g) > 0.4
    grad = torch.randn(M, 4, generator=g)
    state = types.SimpleNamespace(n_offsets=k, opacity_accum=torch.rand(N, 1, generator=g), anchor_demon=torch.rand(N, 1, generator=g).round(),
                                  offset_gradient_accum=torch.rand(N * k, 1, generator=g), offset_denom=torch.rand(N * k, 1, generator=g).round())
    before = {f: getattr(state, f).clone() for f in ("opacity_accum", "anchor_demon", "offset_gradient_accum", "offset_denom")}
    reference_method()(state, types.SimpleNamespace(grad=grad), opacity, update_filter, sel, vis)
    npy = lambda t: t.detach().numpy()
    out.update({f"{tag}_N": N, f"{tag}_k": k, f"{tag}_vis": npy(vis), f"{tag}_opacity": npy(opacity), f"{tag}_sel": npy(sel),
                f"{tag}_update_filter": npy(update_filter), f"{tag}_grad": npy(grad)})
    for f, t in before.items():
        out[f"{tag}_before_{f}"] = npy(t); out[f"{tag}_after_{f}"] = npy(getattr(state, f))


if __name__ == "__main__":
    out = {}
    run("a", 500, 6, 1, out)
    run("b", 333, 10, 2, out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
