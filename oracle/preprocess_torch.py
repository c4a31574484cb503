"""Baseline B3: the rasterizer's per-Gaussian stage K1 (R3/cr/forward.cu:256-384) as vectorised PyTorch ops on
the HOST cores -- what a framework-level CPU implementation of the reference's preprocess costs (`torch.set_num_threads(nproc)`).

TEST / BASELINE INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests/test_oracle_cpu.py); nothing under lidar-gs_amd/ imports
it.  Returns (radii i32[P], tiles_touched i64[P]); checked against oracle/lidargs_oracle.c (the float32 op order differs, so a
Gaussian within an ulp of a ceil/round boundary may differ: > 99.9 % agree)."""
import math

import torch


def preprocess(means3D, scales, rotations, viewmatrix, beams, W, H, scale_modifier=1.0, far=80, near=0):
    f32 = torch.float32
    vm = viewmatrix.reshape(4, 4).to(f32)
    p = means3D.to(f32) @ vm[:3, :3] + vm[3, :3]                                        # transformPoint4x3 (row-vector convention)
    dist = torch.linalg.vector_norm(p, dim=1)
    live = (dist < float(far)) & (dist > float(near))                                   # :304
    q = rotations.to(f32)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
    M = R * (scale_modifier * scales.to(f32)).unsqueeze(1)                               # R S
    Sigma = M @ M.transpose(1, 2)                                                        # :216-253
    d = p / dist.clamp_min(1e-30).unsqueeze(1)
    u1 = torch.stack([d[:, 1], -d[:, 0], torch.zeros_like(dist)], 1)
    n1 = torch.linalg.vector_norm(u1, dim=1, keepdim=True)
    u1 = torch.where(n1 > 0, u1 / n1.clamp_min(1e-30), u1)                                # :95-119
    u2 = torch.linalg.cross(d, u1)
    Rv = vm[:3, :3]                                                                       # world -> view: v_view = v_world @ Rv
    t1, t2 = u1 @ Rv.T, u2 @ Rv.T                                                         # tangents back in world space
    St1, St2 = torch.einsum("pij,pj->pi", Sigma, t1), torch.einsum("pij,pj->pi", Sigma, t2)
    d2 = dist * dist
    a = ((t1 * St1).sum(1) + 0.01) / d2                                                   # :146-169, :319-321
    b = (t1 * St2).sum(1) / d2
    c = ((t2 * St2).sum(1) + 0.01) / d2
    det = a * c - b * b
    live &= det != 0
    mid = 0.5 * (a + c)
    disc = torch.sqrt(torch.clamp((mid * mid - det).double(), min=1e-9))                  # :328-330 (double)
    lam = torch.maximum((mid.double() + disc).float(), (mid.double() - disc).float())
    radius = torch.sqrt(torch.clamp(lam.double(), min=1e-9)).float()
    pi = 3.14159265358979323846
    p_c = (pi - torch.atan2(p[:, 1], p[:, 0])) / (2 * pi / W)                              # :333-334
    alpha = torch.atan2(p[:, 2], torch.sqrt(p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]))       # :336
    beams = beams.to(f32).contiguous()
    i = torch.searchsorted(beams, alpha.contiguous(), right=False)                        # first beam >= alpha (auxiliary.h:41-63)
    i = torch.where(alpha >= beams[-1], torch.full_like(i, H - 1), i)
    i = torch.where(alpha <= beams[0], torch.zeros_like(i), i).clamp_(0, H - 1)
    pos = i > 0
    before = torch.where(pos, beams[(i - 1).clamp_min(0)], beams[0].expand_as(alpha))
    after = torch.where(pos, beams[i], beams[1].expand_as(alpha))
    p_r = torch.where(pos, (i - 1).to(f32) + (alpha - before) / (after - before), 1.0 + (alpha - after) / (after - before))
    live &= torch.where(pos, alpha <= after + 0.004, alpha >= before - 0.004)              # :341-358
    p_r = float(H) - p_r - 1.0
    ry = torch.ceil(3.0 * radius / torch.tan((after - before).abs()))                     # :361-362
    rx = torch.ceil(3.0 * radius / math.tan(2 * pi / W))
    gx = (W + 15) // 16
    xmin = torch.clamp(((p_c - rx) / 16.0).to(torch.int64), 0, gx)                         # auxiliary.h:80-92
    xmax = torch.clamp(((p_c + rx + 15.0) / 16.0).to(torch.int64), 0, gx)
    ymin = torch.clamp(torch.round(p_r - ry).to(torch.int64), 0, H)
    ymax = torch.clamp(torch.maximum(torch.round(p_r + ry), torch.round(p_r) + 1.0).to(torch.int64), 0, H)
    tiles = (xmax - xmin) * (ymax - ymin)
    live &= tiles > 0
    radii = torch.where(live, torch.maximum(rx, ry), torch.zeros_like(rx)).to(torch.int32)
    return radii, torch.where(live, tiles, torch.zeros_like(tiles))
