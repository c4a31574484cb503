"""An oracle check that does not share the restatement's reading of the reference's BACKWARD (round 5; CPU only).

oracle/lidargs_oracle.c restates backward.cu kernel by kernel -- and the HIP kernels were written by the same author from the same
reading, so a misread formula would pass every HIP-vs-oracle test.  Here the reference's FORWARD formulas are written down once more,
as they stand in the source, in float64 torch, and its gradients come from torch.autograd -- nobody's reading of backward.cu is involved:

  blend stage   R3/cr/forward.cu:578-637 (the per-pixel loop: delta, d = (delta.u1/u1.u1, delta.u2/u2.u2), power, alpha, the three skips,
                C / D / T, the background) with K1's per-Gaussian outputs as leaves.  K8 (R3/cr/backward.cu:645-789) claims to be the
                analytic gradient of exactly this, so its sums must equal autograd's up to fp32 rounding:
                dL_dcolors, dL_ddepths, dL_dopacity, dL_dconic (x, w, and y -- which the reference accumulates UN-doubled, :784, and
                doubles back in K9, :247), dL_dbasis_u1 / u2, dL_dsphere_means3D, and dL_dmean2D.xy = the gradient w.r.t. d itself (:753).
                (dL_dmean2D.z is a sum of per-pixel norms, :779 -- a statistic, not a gradient: not checked here.)
  whole chain   K1's formulas (R3/cr/forward.cu:216-253 computeCov3D, :95-119 _proj_2basis, :146-169 computeCov2D_lidar, :298-322,
                :369-372) in front of the blend, leaves = means3D / scales / rotations / opacities / colours.  K9 + K10
                (R3/cr/backward.cu:157-532) must then equal autograd's gradients, with the ONE substitution the reference makes on
                purpose: 1 / denom^2 -> 1 / (denom^2 + 1e-7) in the conic -> covariance step (:237), applied here as a scale on the
                gradient that flows into (a, b, c) -- not as a restated formula.  Its 1e-9 epsilons (:313-354) move results by < 1e-7.

Which Gaussians a pixel's list holds, and in which order, is taken from the oracle's forward state (binning is not differentiable);
every per-pair decision (power > 0, alpha < 1/255, T < 1e-4) is re-taken here in float64, and pairs within 1e-5 of a threshold make
the test fail loudly instead of comparing a flipped pair.  Opacities stay below 0.95 so that min(0.99, .) never clips (a clipped alpha
is the one place where K8 is knowingly not the analytic gradient: it passes the gradient straight through, :676-788).
"""
import math

import numpy as np
import pytest
import torch

import lidargs_scenes as sc
from oracle import lgo

F64 = torch.float64


def _scene(P, H, seed, kind="shell"):
    s = sc.make_scene(kind, P, H, seed, random_view=True)
    s["opacities"] = np.clip(s["opacities"], 0.15, 0.9).astype(np.float32)
    s["bg"] = np.array([0.25, 0.6], np.float32)
    return s


def _pixel_dirs(W, H, beams):
    """R3/cr/forward.cu:589-591: alp = beams[H-1-y], beta = -(x - W/2)/W * 2 pi, (cos a cos b, cos a sin b, sin a)."""
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    alp = torch.as_tensor(beams, dtype=F64)[H - 1 - ys]
    beta = -(xs.to(F64) - W / 2.0) / W * 2.0 * math.pi
    return torch.stack([torch.cos(alp) * torch.cos(beta), torch.cos(alp) * torch.sin(beta), torch.sin(alp)], -1)   # [H, W, 3]


def _blend(lists, dirs, W, H, conic, opac, colors, depth, u1, u2, sph, e, bg, margin_log):
    """The per-pixel loop of renderCUDA, vectorised over a pixel's list.  -> color [2,H,W], depth [H,W], occ [H,W]."""
    out_c = torch.zeros(2, H, W, dtype=F64); out_d = torch.zeros(H, W, dtype=F64); out_o = torch.zeros(H, W, dtype=F64)
    for y in range(H):
        for x in range(W):
            ids = lists(x, y)
            T = torch.ones((), dtype=F64)
            C = torch.zeros(2, dtype=F64); D = torch.zeros((), dtype=F64)
            if len(ids):
                idx = torch.as_tensor(ids, dtype=torch.long)
                delta = sph[idx] - dirs[y, x]                                             # :592
                U1, U2 = u1[idx], u2[idx]
                dx = (delta * U1).sum(-1) / (U1 * U1).sum(-1) + e[idx, 0]                # :593-597 (+ the probe for dL/dd)
                dy = (delta * U2).sum(-1) / (U2 * U2).sum(-1) + e[idx, 1]
                A, B, Cc = conic[idx, 0], conic[idx, 1], conic[idx, 2]
                power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy                # :601
                alpha = opac[idx] * torch.exp(power)                                      # :604 (min(0.99, .) never active: opacities <= 0.9)
                assert float(alpha.detach().max()) < 0.99
                hit = (power <= 0.0) & (alpha >= 1.0 / 255.0)                             # :602, :605
                f = torch.where(hit, 1.0 - alpha, torch.ones_like(alpha))
                T_incl = torch.cumprod(f, 0)
                T_excl = torch.cat([torch.ones(1, dtype=F64), T_incl[:-1]])
                trip = hit & (T_incl < 0.0001)                                            # :607-611: the first one ends the walk
                n_live = int(torch.nonzero(trip)[0]) if bool(trip.any()) else len(ids)
                live = torch.arange(len(ids)) < n_live
                blend = hit & live
                wgt = torch.where(blend, alpha * T_excl, torch.zeros_like(alpha))         # :615-617
                C = (wgt[:, None] * colors[idx]).sum(0); D = (wgt * depth[idx]).sum()
                T = T_excl[n_live] if n_live < len(ids) else T_incl[-1]
                with torch.no_grad():                                                     # distance of every decision from its threshold
                    lv = live.clone(); lv[min(n_live, len(ids) - 1)] = True
                    margin_log.append(float((power.abs()[lv]).min()))
                    margin_log.append(float(((alpha - 1.0 / 255.0).abs() / (1.0 / 255.0))[lv & (power <= 0)].min()) if bool((lv & (power <= 0)).any()) else 1.0)
                    margin_log.append(float(((T_incl - 0.0001).abs() / 0.0001)[lv & hit].min()) if bool((lv & hit).any()) else 1.0)
            out_c[:, y, x] = C + T * bg                                                   # :637
            out_d[y, x] = D; out_o[y, x] = 1.0 - T
    return out_c, out_d, out_o


def _lists_of(f, W, H):
    """point_list / ranges of the oracle's 16 x 1 tiles (R3/cr/rasterizer_impl.cu:117-139) -> ids of pixel (x, y), in blend order."""
    pl = f.array("point_list"); rg = f.array("ranges").reshape(-1, 2)
    tiles_x = (W + 15) // 16
    return lambda x, y: pl[rg[y * tiles_x + x // 16, 0]:rg[y * tiles_x + x // 16, 1]].astype(np.int64)


def _close(name, got, ref, rtol=2e-4, floor=1e-3, worst=3e-3):
    """The oracle (fp32, sums of signed per-pixel terms that cancel a digit or two) against float64 autograd: 99 % of the entries within
    `rtol`, every entry within `worst` (relative to |x| + floor * max|x|).  A misread formula -- a wrong factor, sign or missing term -- moves
    most entries of its array by O(1); fp32 rounding moves the few whose terms cancel, by < 1e-3 (measured: the worst is one Gaussian's
    sphere gradient, 73.63 against 73.74 in an array whose largest entry is 5069)."""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref) / (np.abs(ref) + floor * scale)
    p99 = float(np.quantile(err, 0.99))
    print(f"[autograd] {name:16s} max|x|={scale:.3e}  rel err: p99 {p99:.2e}  worst {err.max():.2e}")
    assert p99 <= rtol, f"{name}: oracle differs from float64 autograd of the reference's forward: p99 {p99:.3e}"
    assert err.max() <= worst, f"{name}: oracle differs from float64 autograd of the reference's forward by {err.max():.3e}"


H_, W_, P_ = 8, 96, 260


@pytest.fixture(scope="module")
def run():
    scene = _scene(P_, H_, 41)
    grads = sc.upstream_grads(H_, W_, 41)
    f = lgo.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"], scene["viewmatrix"],
                    scene["beams"], W_, H_, bg=scene["bg"])
    g = lgo.backward(f, *grads)
    return scene, grads, f, g


def _loss(img, grads):
    gc, gd, go = (torch.as_tensor(np.asarray(x), dtype=F64) for x in grads)
    return (img[0] * gc.reshape(2, H_, W_)).sum() + (img[1] * gd.reshape(H_, W_)).sum() + (img[2] * go.reshape(H_, W_)).sum()


def test_blend_gradients_are_autograd_of_the_forward_loop(run):
    scene, grads, f, g = run
    P = scene["means3D"].shape[0]
    co = f.array("conic_opacity").reshape(P, 4).astype(np.float64)
    leaf = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=F64, requires_grad=True)
    conic = leaf(co[:, :3]); opac = leaf(co[:, 3]); colors = leaf(scene["colors"]); depth = leaf(f.array("depths"))
    u1 = leaf(f.array("basis_u1").reshape(P, 3)); u2 = leaf(f.array("basis_u2").reshape(P, 3)); sph = leaf(f.array("sphere").reshape(P, 3))
    e = torch.zeros(P, 2, dtype=F64, requires_grad=True)
    margins = []
    img = _blend(_lists_of(f, W_, H_), _pixel_dirs(W_, H_, scene["beams"]), W_, H_, conic, opac, colors, depth, u1, u2, sph, e,
                 torch.as_tensor(scene["bg"], dtype=F64), margins)
    assert min(margins) > 1e-5, f"a pair sits within {min(margins):.1e} of a threshold: pick another seed"
    # the float64 forward agrees with the oracle's fp32 images (so the same pairs were blended)
    np.testing.assert_allclose(img[0].detach().numpy(), f.color, rtol=0, atol=2e-5)
    np.testing.assert_allclose(img[1].detach().numpy(), f.depth[0], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(img[2].detach().numpy(), f.occ[0], rtol=0, atol=2e-5)
    gr = torch.autograd.grad(_loss(img, grads), [conic, opac, colors, depth, u1, u2, sph, e])
    gconic, gop, gcol, gdep, gu1, gu2, gs, ge = (t.numpy() for t in gr)
    _close("dL_dcolors", g["dL_dcolors"], gcol)
    _close("dL_ddepths", g["dL_ddepths"][:, 0], gdep)
    _close("dL_dopacity", g["dL_dopacity"][:, 0], gop)
    _close("dL_dconic.x", g["dL_dconic"][:, 0], gconic[:, 0])
    _close("2 dL_dconic.y", 2.0 * g["dL_dconic"][:, 1].astype(np.float64), gconic[:, 1])     # accumulated un-doubled (:784), doubled back in K9 (:247)
    _close("dL_dconic.w", g["dL_dconic"][:, 3], gconic[:, 2])
    _close("dL_dbasis_u1", g["dL_dbasis_u1"], gu1)
    _close("dL_dbasis_u2", g["dL_dbasis_u2"], gu2)
    _close("dL_dsphere", g["dL_dsphere"], gs)
    _close("dL_dmean2D.xy", g["dL_dmeans2D"][:, :2], ge)
    assert np.all(g["dL_dmeans2D"][:, 3] == 0.0)                                             # :780


def _mat3_cols(*c):
    """glm::mat3(a, b, c, d, e, f, g, h, i): consecutive triples are COLUMNS -> math matrix [row, col]."""
    return torch.stack([torch.stack(c[0:3]), torch.stack(c[3:6]), torch.stack(c[6:9])], 1)


def _k1(means3D, scales, rotations, vm, mod=1.0):
    """preprocessCUDA's per-Gaussian outputs for one Gaussian, as written (R3/cr/forward.cu:216-253, :95-119, :146-169, :298-322, :369-372).
    Returns (conic [3], dist, u1 [3], u2 [3], sphere [3], (a, b, c))."""
    p = means3D
    pv = torch.stack([vm[0] * p[0] + vm[4] * p[1] + vm[8] * p[2] + vm[12], vm[1] * p[0] + vm[5] * p[1] + vm[9] * p[2] + vm[13],
                      vm[2] * p[0] + vm[6] * p[1] + vm[10] * p[2] + vm[14]])                   # transformPoint4x3, auxiliary.h:94-102
    dist = torch.sqrt((pv * pv).sum())
    one, zero = torch.ones((), dtype=F64), torch.zeros((), dtype=F64)
    S = torch.diag(torch.stack([mod * scales[0], mod * scales[1], mod * scales[2]]))
    r, x, y, z = rotations
    R = _mat3_cols(1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                   2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                   2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y))
    M = S @ R
    Sigma = M.T @ M                                                                        # :244
    dirv = pv / dist                                                                       # normalize_f3
    u1 = torch.stack([dirv[1], -dirv[0], zero]); u1 = u1 / torch.sqrt((u1 * u1).sum())
    u2 = torch.stack([dirv[1] * u1[2] - dirv[2] * u1[1], dirv[2] * u1[0] - dirv[0] * u1[2], dirv[0] * u1[1] - dirv[1] * u1[0]])
    Pm = _mat3_cols(u1[0], u1[1], u1[2], u2[0], u2[1], u2[2], zero, zero, zero)
    Wm = _mat3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10])
    Tm = Wm @ Pm
    cov = Tm.T @ Sigma.T @ Tm                                                              # :162
    a = (cov[0, 0] + 0.01) / (dist * dist); b = cov[1, 0] / (dist * dist); c = (cov[1, 1] + 0.01) / (dist * dist)   # cov[0][1] = column 0, row 1
    abc = torch.stack([a, b, c])
    return abc, dist, u1, u2, pv / dist, one


def test_whole_chain_gradients_are_autograd_of_k1_and_the_forward_loop(run):
    scene, grads, f, g = run
    P = scene["means3D"].shape[0]
    leaf = lambda a: torch.tensor(np.asarray(a, np.float64), dtype=F64, requires_grad=True)
    means, scales, rots = leaf(scene["means3D"]), leaf(scene["scales"]), leaf(scene["rotations"])
    opac, colors = leaf(scene["opacities"][:, 0]), leaf(scene["colors"])
    vm = torch.as_tensor(np.asarray(scene["viewmatrix"], np.float64).reshape(16), dtype=F64)
    vis = f.radii > 0
    conic, depth, u1, u2, sph = [], [], [], [], []
    z3 = torch.zeros(3, dtype=F64)
    for i in range(P):
        if not vis[i]:                                                                     # culled: in nobody's list
            conic.append(z3); depth.append(torch.zeros((), dtype=F64)); u1.append(z3); u2.append(z3); sph.append(z3)
            continue
        abc, dist, a1, a2, s, _ = _k1(means[i], scales[i], rots[i], vm)
        det = abc[0] * abc[2] - abc[1] * abc[1]
        # THE substitution (R3/cr/backward.cu:237): the gradient that reaches (a, b, c) through the conic's 1 / det^2 is formed with
        # 1 / (det^2 + 1e-7) instead -- i.e. scaled by det^2 / (det^2 + 1e-7)
        k = (det * det / (det * det + 1e-7)).detach()
        abc_r = abc * k + (abc * (1 - k)).detach()
        ar, br, cr = abc_r
        detr = ar * cr - br * br
        conic.append(torch.stack([cr / detr, -br / detr, ar / detr]))                      # :322
        depth.append(dist); u1.append(a1); u2.append(a2); sph.append(s)
    conic, depth, u1, u2, sph = torch.stack(conic), torch.stack(depth), torch.stack(u1), torch.stack(u2), torch.stack(sph)
    # K1 itself agrees with the oracle's state (fp32) before anything is differentiated
    co = f.array("conic_opacity").reshape(P, 4)
    np.testing.assert_allclose(conic.detach().numpy()[vis], co[vis, :3], rtol=3e-4)
    np.testing.assert_allclose(u1.detach().numpy()[vis], f.array("basis_u1").reshape(P, 3)[vis], atol=2e-6)
    np.testing.assert_allclose(u2.detach().numpy()[vis], f.array("basis_u2").reshape(P, 3)[vis], atol=2e-6)
    e = torch.zeros(P, 2, dtype=F64)
    margins = []
    img = _blend(_lists_of(f, W_, H_), _pixel_dirs(W_, H_, scene["beams"]), W_, H_, conic, opac, colors, depth, u1, u2, sph, e,
                 torch.as_tensor(scene["bg"], dtype=F64), margins)
    assert min(margins) > 1e-5
    gm, gsc, gq, gop, gcol = (t.numpy() for t in torch.autograd.grad(_loss(img, grads), [means, scales, rots, opac, colors]))
    _close("dL_dmeans3D", g["dL_dmeans3D"], gm)
    _close("dL_dopacity", g["dL_dopacity"][:, 0], gop)
    _close("dL_dcolors", g["dL_dcolors"], gcol)
    # the scale and rotation gradients pass through the damped conic -> covariance step: five orders of magnitude below the means'
    # (SURVEY.md App. D), still held to the same relative bar
    _close("dL_dscales", g["dL_dscales"], gsc)
    _close("dL_drotations", g["dL_drotations"], gq)
