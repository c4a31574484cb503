"""Generates tests/golden/neural_gaussians_golden.npz by EXECUTING the reference's own `generate_neural_gaussians`
(/root/reference/gaussian_renderer/__init__.py:17-119) on CPU torch, forward and autograd backward.

The function's source is read from the reference checkout at run time and executed (nothing of it is stored here); its
module cannot be imported as a whole in this image (torch_scatter, simple_knn, plyfile and a CUDA device are missing), so
the one function is taken out of the parsed module by name.  The `pc` it receives is a plain object carrying the tensors
and the four MLPs built exactly as GaussianModel.__init__ declares them (scene/gaussian_model.py:113-142, minus .cuda()).

    python tests/golden/make_neural_gaussians_golden.py
"""
import ast
import os
import types

import numpy as np
import torch
from einops import repeat
from torch import nn

REF = "/root/reference/gaussian_renderer/__init__.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural_gaussians_golden.npz")


def reference_function():
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "generate_neural_gaussians")
    fn.args.args[1].annotation = None                      # `pc : GaussianModel` -- the class cannot be imported here
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "repeat": repeat}
    exec(compile(ast.fix_missing_locations(mod), REF, "exec"), ns)
    return ns["generate_neural_gaussians"]


def build_pc(N, k, seed, add_opacity_dist, add_cov_dist, add_color_dist):
    g = torch.Generator().manual_seed(seed)
    feat_dim, hidden = 32, 32
    pc = types.SimpleNamespace()
    pc.use_feat_bank, pc.appearance_dim, pc.n_offsets, pc.color_channel = False, 0, k, 2
    pc.add_opacity_dist, pc.add_cov_dist, pc.add_color_dist = add_opacity_dist, add_cov_dist, add_color_dist
    torch.manual_seed(seed)
    mk = lambda din, dout, act: nn.Sequential(nn.Linear(din, hidden), nn.ReLU(True), nn.Linear(hidden, dout), *([act] if act else []))
    pc.mlp_opacity = mk(feat_dim + 3 + int(add_opacity_dist), k, nn.Tanh())
    pc.mlp_cov = mk(feat_dim + 3 + int(add_cov_dist), 7 * k, None)
    pc.mlp_color = mk(feat_dim + 3 + int(add_color_dist), (pc.color_channel - 1) * k, nn.Sigmoid())
    pc.mlp_raydrop = mk(feat_dim + 3 + int(add_color_dist), k, nn.Sigmoid())
    pc.get_opacity_mlp, pc.get_cov_mlp, pc.get_color_mlp, pc.get_raydrop_mlp = pc.mlp_opacity, pc.mlp_cov, pc.mlp_color, pc.mlp_raydrop
    pc._anchor_feat = (torch.randn(N, feat_dim, generator=g) * 0.5).requires_grad_(True)
    pc._anchor = (torch.randn(N, 3, generator=g) * 10.0).requires_grad_(True)
    pc.get_anchor = pc._anchor
    pc._offset = (torch.randn(N, k, 3, generator=g) * 0.3).requires_grad_(True)
    pc._scaling = (torch.randn(N, 6, generator=g) * 0.3 - 1.0).requires_grad_(True)
    pc.get_scaling_leaf = torch.exp(pc._scaling).detach().requires_grad_(True)       # get_scaling = 1.0 * exp(_scaling), :213-214
    pc.get_scaling = pc.get_scaling_leaf
    pc.rotation_activation = torch.nn.functional.normalize                           # gaussian_model.py:47
    return pc, g


def run(tag, N, k, seed, flags, out):
    fn = reference_function()
    pc, g = build_pc(N, k, seed, *flags)
    cam = types.SimpleNamespace(camera_center=torch.tensor([0.3, -0.2, 1.1]), uid=0)
    vis = torch.rand(N, generator=g) > 0.2
    xyz, color, opacity, scaling, rot, neural_opacity, mask = fn(cam, pc, vis, is_training=True)
    ups = [torch.randn(t.shape, generator=g) for t in (xyz, color, opacity, scaling, rot)]
    loss = sum((u * t).sum() for u, t in zip(ups, (xyz, color, opacity, scaling, rot)))
    loss.backward()
    npy = lambda t: t.detach().numpy().astype(np.float32)
    out.update({f"{tag}_N": N, f"{tag}_k": k, f"{tag}_flags": np.array(flags), f"{tag}_cam": npy(cam.camera_center), f"{tag}_vis": vis.numpy(),
                f"{tag}_anchor_feat": npy(pc._anchor_feat), f"{tag}_anchor": npy(pc._anchor), f"{tag}_offset": npy(pc._offset),
                f"{tag}_scaling_in": npy(pc.get_scaling_leaf)})
    for name, mlp in (("opacity", pc.mlp_opacity), ("cov", pc.mlp_cov), ("color", pc.mlp_color), ("raydrop", pc.mlp_raydrop)):
        out[f"{tag}_{name}_W1"], out[f"{tag}_{name}_b1"] = npy(mlp[0].weight), npy(mlp[0].bias)
        out[f"{tag}_{name}_W2"], out[f"{tag}_{name}_b2"] = npy(mlp[2].weight), npy(mlp[2].bias)
        out[f"{tag}_g_{name}_W1"], out[f"{tag}_g_{name}_b1"] = npy(mlp[0].weight.grad), npy(mlp[0].bias.grad)
        out[f"{tag}_g_{name}_W2"], out[f"{tag}_g_{name}_b2"] = npy(mlp[2].weight.grad), npy(mlp[2].bias.grad)
    for name, t in (("xyz", xyz), ("color", color), ("opacity", opacity), ("scaling", scaling), ("rot", rot), ("neural_opacity", neural_opacity)):
        out[f"{tag}_out_{name}"] = npy(t)
    out[f"{tag}_out_mask"] = mask.numpy()
    for name, u in zip(("xyz", "color", "opacity", "scaling", "rot"), ups):
        out[f"{tag}_up_{name}"] = npy(u)
    out[f"{tag}_g_anchor_feat"], out[f"{tag}_g_anchor"] = npy(pc._anchor_feat.grad), npy(pc._anchor.grad)
    out[f"{tag}_g_offset"], out[f"{tag}_g_scaling"] = npy(pc._offset.grad), npy(pc.get_scaling_leaf.grad)


if __name__ == "__main__":
    out = {}
    run("a", 400, 6, 1, (True, True, True), out)          # the reference's default configuration (arguments/__init__.py:53,77-79)
    run("b", 300, 5, 2, (False, True, False), out)        # mixed dist flags, another offset count
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
