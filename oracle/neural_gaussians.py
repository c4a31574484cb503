"""CPU restatement (numpy, fp32) of LiDAR-GS's anchor decode `generate_neural_gaussians`
(/root/reference/gaussian_renderer/__init__.py:17-119) and of the gradients torch autograd derives for it.

TEST INFRASTRUCTURE ONLY (same rules as lidargs_oracle.c).  PARITY STATUS: pinned -- tests/golden/neural_gaussians_golden.npz holds
inputs, outputs and autograd gradients produced by EXECUTING the reference's own function on CPU torch
(tests/golden/make_neural_gaussians_golden.py); tests/test_neural_gaussians_cpu.py checks this file against them.

Model configuration restated (arguments/__init__.py:51-79, scene/gaussian_model.py:113-142): feat_dim 32, hidden 32,
k = n_offsets, appearance_dim 0, use_feat_bank False, add_{opacity,cov,color}_dist independently on/off, colour
channels 2 (intensity MLP with (2-1)*k outputs + ray-drop MLP with k outputs).
The four MLPs are Linear-ReLU-Linear with output activations tanh / identity / sigmoid / sigmoid.
"""
import numpy as np

F32 = np.float32
MLPS = ("opacity", "cov", "color", "raydrop")


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


def _mlp_forward(x, W1, b1, W2, b2):
    pre = (x @ W1.T + b1).astype(F32)          # nn.Linear: y = x W^T + b  (gaussian_model.py:115-142)
    h = np.maximum(pre, 0).astype(F32)
    return h, (h @ W2.T + b2).astype(F32)


def forward(p, cam_center, visible_mask=None):
    """p: dict(anchor_feat [N,32], anchor [N,3], offset [N,k,3], scaling [N,6] (already exp-activated, get_scaling),
    {mlp}_W1/_b1/_W2/_b2, add_opacity_dist/add_cov_dist/add_color_dist).  Returns the 7-tuple of the training path as a dict
    plus the intermediates the backward needs.  Line numbers: gaussian_renderer/__init__.py."""
    N = p["anchor"].shape[0]
    vis = np.ones(N, bool) if visible_mask is None else np.asarray(visible_mask, bool)          # :19-20
    feat, anchor = p["anchor_feat"][vis].astype(F32), p["anchor"][vis].astype(F32)              # :22-23
    offs, scal = p["offset"][vis].astype(F32), p["scaling"][vis].astype(F32)                    # :24-25
    n, k = anchor.shape[0], offs.shape[1]
    ob = (anchor - np.asarray(cam_center, F32)).astype(F32)                                     # :28
    dist = np.sqrt((ob * ob).sum(1, keepdims=True, dtype=F32)).astype(F32)                      # :32
    view = (ob / dist).astype(F32)                                                              # :34
    x_d = np.concatenate([feat, view, dist], 1).astype(F32)                                     # :50
    x_nd = x_d[:, :-1]                                                                          # :51
    xin = {"opacity": x_d if p["add_opacity_dist"] else x_nd, "cov": x_d if p["add_cov_dist"] else x_nd,
           "color": x_d if p["add_color_dist"] else x_nd, "raydrop": x_d if p["add_color_dist"] else x_nd}
    hid, out = {}, {}
    for m in MLPS:
        hid[m], out[m] = _mlp_forward(xin[m], p[m + "_W1"], p[m + "_b1"], p[m + "_W2"], p[m + "_b2"])
    neural_opacity = np.tanh(out["opacity"]).astype(F32).reshape(-1, 1)                         # :60-66 (Tanh is part of the MLP)
    mask = (neural_opacity > 0.0).reshape(-1)                                                   # :67-68
    color = _sigmoid(out["color"]).reshape(n * k, 1)                                            # :85
    raydrop = _sigmoid(out["raydrop"]).reshape(n * k, 1)                                        # :86
    color2 = np.concatenate([color, raydrop], 1)                                                # :87
    scale_rot = out["cov"].reshape(n * k, 7)                                                    # :94
    offsets = offs.reshape(-1, 3)                                                               # :97
    rep = np.repeat(np.concatenate([scal, anchor], 1), k, axis=0)                               # :100-101
    sel = np.nonzero(mask)[0]
    scaling_rep, anchor_rep = rep[sel, :6], rep[sel, 6:9]                                       # :103-104
    sr = scale_rot[sel]
    scaling = (scaling_rep[:, 3:] * _sigmoid(sr[:, :3])).astype(F32)                           # :107
    q = sr[:, 3:7]
    qn = np.maximum(np.sqrt((q * q).sum(1, keepdims=True, dtype=F32)), F32(1e-12)).astype(F32)  # F.normalize eps
    rot = (q / qn).astype(F32)                                                                  # :108
    off_m = (offsets[sel] * scaling_rep[:, :3]).astype(F32)                                     # :111
    xyz = (anchor_rep + off_m).astype(F32)                                                      # :112
    return dict(xyz=xyz, color=color2[sel], opacity=neural_opacity[sel], scaling=scaling, rot=rot,
                neural_opacity=neural_opacity, mask=mask,
                _ctx=dict(vis=vis, n=n, k=k, xin=xin, hid=hid, out=out, sel=sel, view=view, dist=dist, ob=ob, scal=scal, anchor=anchor,
                          offsets=offsets, q=q, qn=qn, sr=sr, x_d=x_d))


def backward(p, fwd, g_xyz, g_color, g_opacity, g_scaling, g_rot):
    """Gradients of sum(g . output) w.r.t. anchor_feat, anchor, offset, scaling and the 16 MLP tensors: what torch autograd
    returns for the reference function (checked against the golden fixture)."""
    c = fwd["_ctx"]
    n, k, sel = c["n"], c["k"], c["sel"]
    N = p["anchor"].shape[0]
    f64 = np.float64
    nk = n * k
    # --- per-(anchor, offset) output-layer deltas, zero where the offset is masked out
    d_out = {m: np.zeros((nk, w), f64) for m, w in (("opacity", 1), ("color", 1), ("raydrop", 1), ("cov", 7))}
    o = fwd["neural_opacity"][sel].astype(f64)
    d_out["opacity"][sel] = g_opacity.astype(f64) * (1.0 - o * o)                               # tanh'
    col = fwd["color"].astype(f64)
    d_out["color"][sel] = g_color[:, :1].astype(f64) * col[:, :1] * (1.0 - col[:, :1])         # sigmoid'
    d_out["raydrop"][sel] = g_color[:, 1:2].astype(f64) * col[:, 1:2] * (1.0 - col[:, 1:2])
    sr = c["sr"].astype(f64)
    sg = 1.0 / (1.0 + np.exp(-sr[:, :3]))
    rep_scal = np.repeat(c["scal"].astype(f64), k, axis=0)[sel]
    d_sr = np.zeros((sel.size, 7), f64)
    d_sr[:, :3] = g_scaling.astype(f64) * rep_scal[:, 3:] * sg * (1.0 - sg)
    q, qn = c["q"].astype(f64), c["qn"].astype(f64)
    rot = q / qn
    gr = g_rot.astype(f64)
    d_sr[:, 3:7] = (gr - rot * (gr * rot).sum(1, keepdims=True)) / qn                           # normalize VJP (|q| > eps)
    d_out["cov"][sel] = d_sr
    # --- direct paths: xyz = anchor + offset * scaling[:3];  scaling_out = scaling[3:] * sigmoid(.)
    g_off_m = g_xyz.astype(f64)
    d_offsets = np.zeros((nk, 3), f64); d_offsets[sel] = g_off_m * rep_scal[:, :3]
    d_scal_rep = np.zeros((nk, 6), f64)
    d_scal_rep[sel, :3] = g_off_m * c["offsets"].astype(f64)[sel]
    d_scal_rep[sel, 3:] = g_scaling.astype(f64) * sg
    d_anchor_rep = np.zeros((nk, 3), f64); d_anchor_rep[sel] = g_off_m
    d_scal = d_scal_rep.reshape(n, k, 6).sum(1)
    d_anchor = d_anchor_rep.reshape(n, k, 3).sum(1)
    # --- MLPs
    grads = {}
    d_x = np.zeros((n, 36), f64)
    for m, width in (("opacity", 1), ("cov", 7), ("color", 1), ("raydrop", 1)):
        d2 = d_out[m].reshape(n, k * width)
        h, x = c["hid"][m].astype(f64), c["xin"][m].astype(f64)
        W1, W2 = p[m + "_W1"].astype(f64), p[m + "_W2"].astype(f64)
        grads[m + "_W2"] = d2.T @ h
        grads[m + "_b2"] = d2.sum(0)
        d1 = (d2 @ W2) * (h > 0)
        grads[m + "_W1"] = d1.T @ x
        grads[m + "_b1"] = d1.sum(0)
        dx = d1 @ W1
        d_x[:, :dx.shape[1]] += dx
    d_feat = d_x[:, :32]
    d_view, d_dist = d_x[:, 32:35], d_x[:, 35:36]
    view, dist = c["view"].astype(f64), c["dist"].astype(f64)
    # view = ob / dist, dist = |ob|
    d_ob = d_view / dist - view * ((d_view * view).sum(1, keepdims=True) / dist) + d_dist * view
    d_anchor = d_anchor + d_ob
    vis = c["vis"]
    full = lambda a, shape: (lambda z: (z.__setitem__(vis, a.reshape((n,) + shape[1:])), z)[1])(np.zeros(shape, f64))
    out = dict(anchor_feat=full(d_feat, (N, 32)), anchor=full(d_anchor, (N, 3)), offset=full(d_offsets.reshape(n, k, 3), (N, k, 3)),
               scaling=full(d_scal, (N, 6)))
    out.update(grads)
    return {key: v.astype(F32) for key, v in out.items()}
