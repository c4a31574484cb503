"""The anchor decode as the reference writes it -- a chain of framework ops -- restated with torch so that it can run on any
device.  TEST / MEASUREMENT INFRASTRUCTURE ONLY: it is the "what the reference's graph costs on this GPU" leg of
tools/time_decode.py and bench.py --workload decode; the product never imports it.  Follows gaussian_renderer/__init__.py:17-119
for the supported configuration (no feature bank, no appearance embedding, 2 colour channels)."""
import torch
import torch.nn.functional as F


def mlp(x, W1, b1, W2, b2, act):
    y = F.linear(F.relu(F.linear(x, W1, b1)), W2, b2)
    return act(y) if act is not None else y


def generate(anchor_feat, anchor, offset, scaling, params, cam, visible_mask, flags):
    """params: dict name -> (W1, b1, W2, b2); returns the 7-tuple of the training path."""
    if visible_mask is None:
        visible_mask = torch.ones(anchor.shape[0], dtype=torch.bool, device=anchor.device)
    feat, anc, offs, scal = anchor_feat[visible_mask], anchor[visible_mask], offset[visible_mask], scaling[visible_mask]   # :22-25
    k = offs.shape[1]
    ob = anc - cam                                                        # :28
    dist = ob.norm(dim=1, keepdim=True)                                   # :32
    view = ob / dist                                                      # :34
    x_d = torch.cat([feat, view, dist], dim=1)                            # :50
    x_nd = torch.cat([feat, view], dim=1)                                 # :51
    pick = lambda f: x_d if f else x_nd
    neural_opacity = mlp(pick(flags[0]), *params["opacity"], torch.tanh).reshape(-1, 1)          # :60-66
    mask = (neural_opacity > 0.0).view(-1)                                # :67-68
    opacity = neural_opacity[mask]                                        # :71
    color = mlp(pick(flags[2]), *params["color"], torch.sigmoid).reshape(anc.shape[0] * k, 1)     # :76-85
    raydrop = mlp(pick(flags[2]), *params["raydrop"], torch.sigmoid).reshape(anc.shape[0] * k, 1)
    color = torch.cat([color, raydrop], dim=1)                            # :87
    scale_rot = mlp(pick(flags[1]), *params["cov"], None).reshape(anc.shape[0] * k, 7)            # :90-94
    offsets = offs.reshape(-1, 3)                                         # :97
    rep = torch.cat([scal, anc], dim=-1).repeat_interleave(k, dim=0)      # :100-101
    masked = torch.cat([rep, color, scale_rot, offsets], dim=-1)[mask]    # :102-103
    scaling_repeat, repeat_anchor, color, scale_rot, offsets = masked.split([6, 3, 2, 7, 3], dim=-1)
    scaling_out = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])  # :107
    rot = F.normalize(scale_rot[:, 3:7])                                  # :108
    xyz = repeat_anchor + offsets * scaling_repeat[:, :3]                 # :111-112
    return xyz, color, opacity, scaling_out, rot, neural_opacity, mask
