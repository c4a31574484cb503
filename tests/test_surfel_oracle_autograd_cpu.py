"""The surfel oracle's backward against float64 autograd of the reference's surfel FORWARD (round 5; CPU only) -- the twin of
tests/test_oracle_autograd_cpu.py for `oracle/lidargs_surfel_oracle.c`, the one oracle that nothing else pins.

R2's backward is NOT the plain gradient of its forward; it is the gradient with a stated set of detachments and two heuristics, all
visible in the source and mirrored here as detachments of the float64 forward, never as restated backward formulas:
  * only colour channel 0 reaches alpha (R2/cr/backward.cu:358-359: `if (ch == 0)`); both channels reach the colours and, through
    T_final x bg, the alphas (:404-407);
  * DETACH_WEIGHT = 1 (:375-381): the distortion plane differentiates through the mapped depths m only, its blend weights are constants;
  * the median depth is a selection (:370-373): its gradient goes to the depth of the selected entry;
  * a pair whose alpha is clipped at 0.99 passes the gradient straight through (never active here: opacities <= 0.9);
  * 2-D branch (rho2d < rho3d, :578-599): the projected centre (x, y) is a function of T_w that the backward replaces by a linearisation
    on the AVERAGE beam spacing (:425, :590-598).  Here (x, y) is a leaf; autograd's dL/d(x, y), pushed through that stated linearisation
    (five per-surfel factors), is added to autograd's dL/dT_w before the comparison;
  * dL_dmean2D (:564-577, :582-585) and dL_dtransMat_2dtemp (:555-557) are sums of ABSOLUTE per-pixel terms -- densification statistics,
    not gradients: not checked;
  * K10''s rotation gradient is the gradient w.r.t. the NORMALISED quaternion (quat_to_rotmat_vjp, R2/cr/auxiliary.h:273-316, stops at
    w, x, y, z = quat * rsqrt(|quat|^2)): the normalisation's Jacobian is not applied, so here the normalised quaternion is the leaf.
Checked: dL_dcolors, dL_dopacity, dL_dnormal, all nine entries of dL_dtransMat (blend stage, K7' forward R2/cr/forward.cu:420-545 with
K1''s outputs as leaves), and K10' (R2/cr/backward.cu:607-700) as the vector-Jacobian product of K1''s transMat / normal formulas
(R2/cr/forward.cu:269-302, R2/cr/auxiliary.h:249-271) -> dL_dmeans3D, dL_dscales, dL_drotations."""
import math

import numpy as np
import pytest
import torch

from oracle import lgo_surfel
from util import surfel_scene, surfel_upstream_grads

F64 = torch.float64
H_, W_, P_ = 8, 96, 220
NEAR_N, FAR_N = 0.2, 80.0


@pytest.fixture(scope="module")
def run():
    scene = surfel_scene("shell", P_, H_, 43, random_view=True)
    scene["opacities"] = np.clip(scene["opacities"], 0.15, 0.9).astype(np.float32)
    scene["bg"] = np.array([0.2, 0.5], np.float32)
    grads = surfel_upstream_grads(H_, W_, 43)
    f = lgo_surfel.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"], scene["viewmatrix"],
                           scene["beams"], W_, H_, bg=scene["bg"])
    g = lgo_surfel.backward(f, *grads)
    return scene, grads, f, g


def _close(name, got, ref, rtol=2e-4, floor=1e-3, worst=1e-2):
    """99 % of the entries within `rtol`, every entry within `worst` (see tests/test_oracle_autograd_cpu.py _close).  The surfel's per-pair
    terms go through the hit point dp = real_depth * p - T_w, which cancels three digits in fp32, and a surfel's sum over its pixels cancels
    again: single entries of the fp32 oracle sit up to 5e-3 from the float64 value while p99 stays below 5e-5 -- rounding, not a formula."""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref) / (np.abs(ref) + floor * scale)
    p99 = float(np.quantile(err, 0.99))
    print(f"[surfel autograd] {name:18s} max|x|={scale:.3e}  rel err: p99 {p99:.2e}  worst {err.max():.2e}")
    assert p99 <= rtol, f"{name}: oracle differs from float64 autograd of the reference's forward: p99 {p99:.3e}"
    assert err.max() <= worst, f"{name}: oracle differs from float64 autograd of the reference's forward by {err.max():.3e}"


def _blend(f, scene, Tu, Tv, Tw, nrm, opac, colors, xy, margins):
    """renderCUDA of R2 (forward.cu:420-545), vectorised over a pixel's list; the detachments of the module docstring applied."""
    pl = f.array("point_list"); rg = f.array("ranges").reshape(-1, 2)
    tiles_x = (W_ + 15) // 16
    beams = torch.as_tensor(scene["beams"], dtype=F64)
    bg = torch.as_tensor(scene["bg"], dtype=F64)
    color = torch.zeros(2, H_, W_, dtype=F64); others = torch.zeros(7, H_, W_, dtype=F64)
    for y in range(H_):
        for x in range(W_):
            t = y * tiles_x + x // 16
            ids = pl[rg[t, 0]:rg[t, 1]].astype(np.int64)
            T = torch.ones((), dtype=F64)
            C = torch.zeros(2, dtype=F64); D = torch.zeros((), dtype=F64); N = torch.zeros(3, dtype=F64)
            med = torch.zeros((), dtype=F64); dist = torch.zeros((), dtype=F64)
            if len(ids):
                idx = torch.as_tensor(ids, dtype=torch.long)
                beta = -(float(x) - W_ / 2.0) / W_ * 2.0 * math.pi                        # :433
                alp = beams[H_ - 1 - y]
                p = torch.stack([torch.cos(alp) * math.cos(beta), torch.cos(alp) * math.sin(beta), torch.sin(alp)])   # :440-444
                tu, tv, tw, n = Tu[idx], Tv[idx], Tw[idx], nrm[idx]
                rho_r = torch.sqrt((tw * tw).sum(-1))                                       # :436
                lam = (tw * n).sum(-1)                                                      # :449-451: L2_Tw cos_phi1
                cphi2 = (n * p).sum(-1)                                                     # :452
                real_depth = lam / cphi2                                                    # :455-458
                dp = real_depth[:, None] * p - tw                                           # :459-460
                sx = (dp * tu).sum(-1) / (tu * tu).sum(-1); sy = (dp * tv).sum(-1) / (tv * tv).sum(-1)   # :461-465
                rho3d = sx * sx + sy * sy
                dx = xy[idx, 0] - float(x); dy = xy[idx, 1] - float(y)                      # :467
                rho2d = 2.0 * (40.0 * dx * dx + 100.0 * dy * dy)                            # :468 (FilterInvSquare = 2)
                in3d = (rho3d <= rho2d) & (real_depth > 0)
                rho = torch.where(real_depth > 0, torch.minimum(rho3d, rho2d), rho2d)       # :471
                depth = torch.where(in3d, real_depth, rho_r)                                # :472
                power = -0.5 * rho
                alpha = opac[idx] * torch.exp(power)                                        # :483 (never clipped: opacities <= 0.9)
                ok = (cphi2 != 0) & (depth >= NEAR_N) & (power <= 0) & (alpha >= 1.0 / 255.0)   # :453, :473, :476, :484
                fct = torch.where(ok, 1.0 - alpha, torch.ones_like(alpha))
                T_incl = torch.cumprod(fct, 0)
                T_excl = torch.cat([torch.ones(1, dtype=F64), T_incl[:-1]])
                trip = ok & (T_incl < 0.0001)                                               # :486-490
                n_live = int(torch.nonzero(trip)[0]) if bool(trip.any()) else len(ids)
                live = torch.arange(len(ids)) < n_live
                bl = ok & live
                w = torch.where(bl, alpha * T_excl, torch.zeros_like(alpha))                # :492
                wd = w.detach()
                m = FAR_N / (FAR_N - NEAR_N) * (1.0 - NEAR_N / depth)                       # :497
                # distortion (:498-501) = sum_i w_i (m_i^2 A_<i + M2_<i - 2 m_i M1_<i), weights detached (DETACH_WEIGHT)
                A_ex = torch.cumsum(wd, 0) - wd; M1_ex = torch.cumsum(wd * m, 0) - wd * m; M2_ex = torch.cumsum(wd * m * m, 0) - wd * m * m
                # (A = 1 - T before the entry: with T_excl = 1 - sum of the weights in front of it)
                dist = (wd * (m * m * A_ex + M2_ex - 2.0 * m * M1_ex)).sum()
                D = (w * depth).sum()
                sel = bl & (T_excl > 0.5)                                                   # :503-507: the last blended entry that saw T > 0.5
                if bool(sel.any()):
                    med = depth[int(torch.nonzero(sel)[-1])]
                N = (w[:, None] * n).sum(0)
                C = torch.stack([(w * colors[idx, 0]).sum(), (wd * colors[idx, 1]).sum()])  # channel 1 does not reach alpha
                T = T_excl[n_live] if n_live < len(ids) else T_incl[-1]
                with torch.no_grad():
                    lv = live.clone(); lv[min(n_live, len(ids) - 1)] = True
                    margins.append(float(((alpha - 1.0 / 255.0).abs() / (1.0 / 255.0))[lv].min()))
                    margins.append(float(((T_incl - 0.0001).abs() / 0.0001)[lv & ok].min()) if bool((lv & ok).any()) else 1.0)
                    margins.append(float(((rho3d - rho2d).abs() / (rho2d.abs() + 1e-12))[lv].min()))
                    margins.append(float((T_excl - 0.5).abs()[lv].min()))
            color[:, y, x] = C + T * bg
            others[0, y, x] = D; others[1, y, x] = 1.0 - T; others[2:5, y, x] = N; others[5, y, x] = med; others[6, y, x] = dist
    return color, others


def test_surfel_blend_gradients_are_autograd_of_the_forward_loop(run):
    scene, grads, f, g = run
    P = P_
    tm = f.array("transMat").reshape(P, 9).astype(np.float64)
    no = f.array("normal_opacity").reshape(P, 4).astype(np.float64)
    leaf = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=F64, requires_grad=True)
    Tu, Tv, Tw, nrm = leaf(tm[:, 0:3]), leaf(tm[:, 3:6]), leaf(tm[:, 6:9]), leaf(no[:, :3])
    opac, colors, xy = leaf(no[:, 3]), leaf(scene["colors"]), leaf(f.array("means2D").reshape(P, 2))
    margins = []
    color, others = _blend(f, scene, Tu, Tv, Tw, nrm, opac, colors, xy, margins)
    assert min(margins) > 1e-5, f"a pair sits within {min(margins):.1e} of a threshold: pick another seed"
    np.testing.assert_allclose(color.detach().numpy(), f.color, rtol=0, atol=3e-5)
    for k in (0, 1, 2, 3, 4, 5):
        np.testing.assert_allclose(others[k].detach().numpy(), f.others[k], rtol=3e-5, atol=3e-4)
    np.testing.assert_allclose(others[6].detach().numpy(), f.others[6], rtol=1e-3, atol=1e-3)      # distortion: O(1) terms cancelling in fp32
    gc, go = (torch.as_tensor(np.asarray(x), dtype=F64) for x in grads)
    loss = (color * gc.reshape(2, H_, W_)).sum() + (others * go.reshape(7, H_, W_)).sum()
    gTu, gTv, gTw, gn, gop, gcol, gxy = (t.numpy() for t in torch.autograd.grad(loss, [Tu, Tv, Tw, nrm, opac, colors, xy]))
    # the stated linearisation of (x, y)(T_w) on the average beam spacing (R2/cr/backward.cu:425, :590-598), applied to autograd's dL/d(x, y)
    tw = tm[:, 6:9].copy()
    tw[f.radii <= 0] = 1.0                                             # (culled surfels: no pairs, zero gradients; any finite factor)
    rxy = np.sqrt(tw[:, 0] ** 2 + tw[:, 1] ** 2); rr = np.sqrt((tw ** 2).sum(1))
    ga = abs(float(scene["beams"][H_ - 1]) - float(scene["beams"][0])) / (H_ - 1.0)
    ddelx = np.stack([W_ / (2 * math.pi) * tw[:, 1] / rxy ** 2, -W_ / (2 * math.pi) * tw[:, 0] / rxy ** 2, np.zeros(P)], 1)
    ddely = np.stack([-ga * tw[:, 2] * tw[:, 0] / (rr ** 2 * rxy), -ga * tw[:, 2] * tw[:, 1] / (rr ** 2 * rxy), ga * rxy / rr ** 2], 1)
    gTw_ref = gTw + gxy[:, :1] * ddelx + gxy[:, 1:] * ddely
    gT = g["dL_dtransMat"].reshape(P, 9)
    _close("dL_dcolors", g["dL_dcolors"], gcol)
    _close("dL_dopacity", g["dL_dopacity"][:, 0], gop)
    _close("dL_dnormal", g["dL_dnormal"], gn)
    _close("dL_dtransMat.Tu", gT[:, 0:3], gTu)
    _close("dL_dtransMat.Tv", gT[:, 3:6], gTv)
    _close("dL_dtransMat.Tw", gT[:, 6:9], gTw_ref)


def _mat3_cols(*c):
    return torch.stack([torch.stack(c[0:3]), torch.stack(c[3:6]), torch.stack(c[6:9])], 1)      # glm: consecutive triples are columns


def test_surfel_k10_is_the_vjp_of_k1(run):
    """dL_dmeans3D / dL_dscales / dL_drotations = (dL_dtransMat, dL_dnormal) pulled back through K1''s formulas as written."""
    scene, grads, f, g = run
    P = P_
    leaf = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=F64, requires_grad=True)
    qn = np.asarray(scene["rotations"], np.float64); qn = qn / np.linalg.norm(qn, axis=1, keepdims=True)
    means, scales, rots = leaf(scene["means3D"]), leaf(scene["scales"]), leaf(qn)      # (the NORMALISED quaternion is the leaf: docstring)
    vm = torch.as_tensor(np.asarray(scene["viewmatrix"], np.float64).reshape(16), dtype=F64)
    vis = f.radii > 0
    gT = torch.as_tensor(g["dL_dtransMat"].reshape(P, 9).astype(np.float64)); gN = torch.as_tensor(g["dL_dnormal"].astype(np.float64))
    tm_ref = f.array("transMat").reshape(P, 9); no_ref = f.array("normal_opacity").reshape(P, 4)
    loss = torch.zeros((), dtype=F64)
    Wv = torch.stack([torch.stack([vm[0], vm[4], vm[8]]), torch.stack([vm[1], vm[5], vm[9]]), torch.stack([vm[2], vm[6], vm[10]])])   # rows of the view rotation
    for i in range(P):
        if not vis[i]:
            continue
        w, x, y, z = rots[i]                                                              # auxiliary.h:251-257, already normalised
        R = _mat3_cols(1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y),
                       2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x),
                       2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y))
        L = R @ torch.diag(torch.stack([scales[i, 0], scales[i, 1], torch.ones((), dtype=F64)]))   # forward.cu:271-273 (scale_modifier = 1)
        p = means[i]
        pv = torch.stack([vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12], vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13],
                          vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14]])
        Tu_, Tv_ = Wv @ L[:, 0], Wv @ L[:, 1]                                               # :277-295: rows of T = (world2view L0, world2view L1, p_view)
        n = Wv @ L[:, 2]                                                                    # :275 transformVec4x3
        mult = 1.0 if float(-(pv * n).sum()) > 0 else -1.0                                  # :297-302 DUAL_VISIABLE
        n = mult * n
        T9 = torch.cat([Tu_, Tv_, pv])
        np.testing.assert_allclose(T9.detach().numpy(), tm_ref[i], rtol=2e-5, atol=2e-5)    # K1' itself agrees with the oracle's state
        np.testing.assert_allclose(n.detach().numpy(), no_ref[i, :3], rtol=2e-5, atol=2e-6)
        loss = loss + (T9 * gT[i]).sum() + (n * gN[i]).sum()
    gm, gs, gq = (t.numpy() for t in torch.autograd.grad(loss, [means, scales, rots]))
    _close("dL_dmeans3D", g["dL_dmeans3D"], gm)
    _close("dL_dscales", g["dL_dscales"], gs)
    _close("dL_drotations", g["dL_drotations"], gq)
