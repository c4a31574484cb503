"""Shared helpers of the test-suite: running the HIP path through the drop-in package, running the
oracle, and the parity metric.

Parity metric (north star: "within 1e-4 relative fp32 on identical inputs"):
    err_i = |hip_i - ref_i| / (|ref_i| + FLOOR * max|ref|)         FLOOR = 1e-3
Every entry is put in one of three classes:
    ok     err <= RTOL = 1e-4
    soft   RTOL < err <= SOFT_MAX = 1e-3     rounding that went through a cancellation (s - q loses three digits): at most SOFT_FRAC
                                             of the entries
    flip   err > SOFT_MAX                    a hard threshold of the algorithm landed on the other side (alpha < 1/255, power > 0,
                                             T < 1e-4, ceil / round of the footprint rect): the input sat within an ulp of it and
                                             exp / atan2 / cos round differently (device libm vs host libm vs CUDA libdevice).
                                             A flip moves a pixel by up to a whole contribution and a gradient row by up to a
                                             whole (pixel, Gaussian) term: its COUNT is bounded by FLIP_FRAC of the entries and its
                                             SIZE by FLIP_ABS_MAX x max|ref| -- one contribution cannot exceed the array's largest
                                             entry by much, whereas memory corruption, an overflowed list or a wrong index can and
                                             usually does (round-3 advisor finding: the size used to be unbounded).
(both counts: at least MIN_COUNT = 2, so that a 100-entry array is not judged on a fraction of one entry).  The budgets are what
the suite's GPU run of round 3 used (profiles/r03_i_parity_budget.json: 1.4e-4 soft, 8e-5 flips at the worst call outside the
surfel distortion plane) plus 50 %; soft + flip stays under the 5e-4 of round 2.  One plane has a stated budget of its own: the
surfel variant's distortion (`soft_frac=` below; tests/test_surfel_gpu.py says why).  Other rules are assertions of their own in
the tests that need them (the median depth, a selection, is checked by count).

Budget vs use: conftest.py prints the summary of every run and writes gpurun_out/parity_budget.json (committed per round under
profiles/).  For scale: two conforming evaluations of the reference itself differ by more (tests/test_ulp_band_cpu.py: 0.2-2.5 %
of the gradient entries over 1e-4).
"""
import numpy as np

RTOL = 1e-4
FLOOR = 1e-3
SOFT_MAX = 1e-3
SOFT_FRAC = 2.5e-4
FLIP_FRAC = 1.25e-4
FLIP_ABS_MAX = 0.05     # no entry may be off by more than this times max|ref| (the `scale` of the call); the suite's largest use is 2.6e-3 (profiles/r04_a_parity_budget.json)
MIN_COUNT = 2
# every parity() call of the session, for the "budget used" summary conftest.py prints and writes (gpurun_out/parity_budget.json)
PARITY_LOG = []


def parity(name, hip, ref, rtol=RTOL, floor=FLOOR, verbose=True, scale=None, soft_frac=None):
    hip = np.asarray(hip, dtype=np.float64).ravel()
    ref = np.asarray(ref, dtype=np.float64).ravel()
    assert hip.shape == ref.shape, (name, hip.shape, ref.shape)
    assert np.isfinite(hip).all(), f"{name}: non-finite values in the HIP result"
    if ref.size == 0:
        return dict(name=name, max=0.0, outliers=0, n=0)
    scale = np.abs(ref).max() if scale is None else float(scale)   # `scale`: magnitude of the terms the value is a difference of
    diff = np.abs(hip - ref)
    err = diff / (np.abs(ref) + floor * scale + 1e-30)
    abs_max = float(diff.max()) / (scale + 1e-30)                     # largest difference in units of max|ref|
    soft_max = max(SOFT_MAX, 10.0 * rtol)
    n_soft = int(((err > rtol) & (err <= soft_max)).sum())
    n_flip = int((err > soft_max).sum())
    stats = dict(name=name, max=float(err.max()), p999=float(np.quantile(err, 0.999)), median=float(np.median(err)),
                 outliers=n_soft + n_flip, soft=n_soft, flips=n_flip, n=int(ref.size), scale=float(scale), abs_max=abs_max)
    if verbose:
        print(f"[parity] {name:18s} n={ref.size:9d} scale={scale:.3e} median={stats['median']:.2e} "
              f"p99.9={stats['p999']:.2e} max={stats['max']:.2e} soft(>{rtol:g})={n_soft} flips(>{soft_max:g})={n_flip}")
    allowed_soft = max(MIN_COUNT, int((SOFT_FRAC if soft_frac is None else soft_frac) * ref.size))
    allowed_flip = max(MIN_COUNT, int(FLIP_FRAC * ref.size))
    stats.update(allowed=allowed_soft, allowed_flips=allowed_flip, outlier_frac_used=(n_soft + n_flip) / ref.size, soft_frac_used=n_soft / ref.size,
                 flip_frac_used=n_flip / ref.size, rtol=rtol, soft_max=soft_max)
    PARITY_LOG.append(stats)
    assert n_soft <= allowed_soft, f"{name}: {n_soft} of {ref.size} entries in ({rtol}, {soft_max}] (allowed {allowed_soft}); max err {err.max():.3e}"
    assert n_flip <= allowed_flip, f"{name}: {n_flip} of {ref.size} entries over {soft_max} (threshold flips; allowed {allowed_flip}); max err {err.max():.3e}"
    assert abs_max <= FLIP_ABS_MAX, f"{name}: an entry is off by {abs_max:.3g} x max|ref| (allowed {FLIP_ABS_MAX}): larger than any single contribution"
    return stats


def to_torch(scene, device="cuda"):
    import torch
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in scene.items()}


def make_settings(scene_t, W, H, far=80, near=0, scale_modifier=1.0, debug=False):
    import lidargs_scenes as sc
    return sc.raster_settings(scene_t, W, H, far, near, scale_modifier, debug)


def hip_forward_backward(scene, W, H, grads=None, cov3D_precomp=None, far=80, near=0, scale_modifier=1.0, device="cuda"):
    """Run the product path the way gaussian_renderer.render() does (:168-179) and return numpy results."""
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    st = to_torch(scene, device)
    P = st["means3D"].shape[0]
    leaves = {}
    for k in ("means3D", "colors", "opacities", "scales", "rotations"):
        leaves[k] = st[k].clone().requires_grad_(True)
    means2D = torch.zeros((P, 4), dtype=torch.float32, device=device, requires_grad=True)
    rast = GaussianRasterizer(make_settings(st, W, H, far, near, scale_modifier))
    kw = dict(means3D=leaves["means3D"], means2D=means2D, shs=None, colors_precomp=leaves["colors"], opacities=leaves["opacities"])
    cov_leaf = None
    if cov3D_precomp is not None:
        cov_leaf = torch.from_numpy(cov3D_precomp).to(device).requires_grad_(True)
        kw.update(scales=None, rotations=None, cov3D_precomp=cov_leaf)
    else:
        kw.update(scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
    color, depth, occ, radii = rast(**kw)
    out = dict(color=color.detach().cpu().numpy(), depth=depth.detach().cpu().numpy(), occ=occ.detach().cpu().numpy(),
               radii=radii.cpu().numpy())
    if grads is not None:
        gc, gd, go = (torch.from_numpy(g).to(device) for g in grads)
        torch.autograd.backward([color, depth, occ], [gc, gd, go])
        out.update(dL_dmeans3D=leaves["means3D"].grad.cpu().numpy(), dL_dmeans2D=means2D.grad.cpu().numpy(),
                   dL_dcolors=leaves["colors"].grad.cpu().numpy(), dL_dopacity=leaves["opacities"].grad.cpu().numpy())
        if cov_leaf is not None:
            out.update(dL_dcov3D=cov_leaf.grad.cpu().numpy())
        else:
            out.update(dL_dscales=leaves["scales"].grad.cpu().numpy(), dL_drotations=leaves["rotations"].grad.cpu().numpy())
    return out


def oracle_forward_backward(scene, W, H, grads=None, cov3D_precomp=None, far=80, near=0, scale_modifier=1.0):
    from oracle import lgo
    f = lgo.forward(scene["means3D"], scene["colors"], scene["opacities"],
                    None if cov3D_precomp is not None else scene["scales"],
                    None if cov3D_precomp is not None else scene["rotations"],
                    scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"], scale_modifier=scale_modifier,
                    cov3D_precomp=cov3D_precomp, far=far, near=near)
    out = dict(color=f.color, depth=f.depth, occ=f.occ, radii=f.radii, fwd=f)
    if grads is not None:
        g = lgo.backward(f, *grads)
        out.update(g)
    return out


def oracle_backward_exact_sums(ref, grads, surfel=False, mode=1):
    """The oracle's backward on the forward state of `ref` (what oracle_[surfel_]forward_backward returned) once more, with the
    per-Gaussian sums of the backward blend taken in float64 (lgo_set_accumulate_double / sfo_...: every term stays the float32 value
    the reference computes; its float atomics add them in scheduling order, R3/cr/backward.cu:702-788, this restatement in raster order --
    the float64 sum is the centre both scatter around).  -> dict of the gradient arrays."""
    import ctypes as C
    from oracle import lgo, lgo_surfel
    L = lgo.lib()
    setter = L.sfo_set_accumulate_double if surfel else L.lgo_set_accumulate_double
    setter(C.c_int(mode))      # 2 (diagnostic): the backward's T = T / (1 - alpha) chain in float64 as well
    try:
        return (lgo_surfel if surfel else lgo).backward(ref["fwd"], *grads)
    finally:
        setter(C.c_int(0))


def parity_or_closer(name, hip, ref32, ref64, band=None, rtol=RTOL, floor=FLOOR, max_widths=1.0, **kw):
    """parity(hip, ref32) -- or, where that budget is exceeded, one of two statements about WHY (round-5 verdict item 3):
      closer   the excess is the oracle's own summation error: against the exact (float64) sums of the same float32 terms
               (oracle_backward_exact_sums) HIP has no more entries over rtol than the raster-order float32 oracle itself (+ the flip
               budget: threshold flips are not summation error);
      band     the excess lies in the reference's own band: `band()` -> (lo, hi), the entry-wise envelope of the seven conforming
               evaluations of the reference source (oracle_envelope); every entry where HIP is off by more than rtol is one the oracle
               itself moves by more than rtol / 2, and HIP lies within `max_widths` local widths of the envelope (envelope_residue).
    Either way no entry may be off by more than FLIP_ABS_MAX x max|ref|.  `band` is only called when the first two fail (seven oracle runs)."""
    try:
        return parity(name, hip, ref32, rtol=rtol, floor=floor, **kw)
    except AssertionError as first:
        if PARITY_LOG and PARITY_LOG[-1]["name"] == name:
            PARITY_LOG.pop()                                            # judged below instead
        h = np.asarray(hip, np.float64).ravel(); r32 = np.asarray(ref32, np.float64).ravel(); r64 = np.asarray(ref64, np.float64).ravel()
        scale = np.abs(r64).max()
        den = np.abs(r64) + floor * scale + 1e-30
        e_hip, e_ora = np.abs(h - r64) / den, np.abs(r32 - r64) / den
        n_hip, n_ora = int((e_hip > rtol).sum()), int((e_ora > rtol).sum())
        allowed_flip = max(MIN_COUNT, int(FLIP_FRAC * r64.size))
        print(f"[parity] {name:18s} vs exact sums: HIP over {rtol:g}: {n_hip}, float32 raster-order oracle over: {n_ora} (n={r64.size}); "
              f"HIP p99.9 {np.quantile(e_hip, 0.999):.2e} max {e_hip.max():.2e}, oracle p99.9 {np.quantile(e_ora, 0.999):.2e} max {e_ora.max():.2e}")
        entry = dict(name=name + " (vs exact sums)", max=float(e_hip.max()), p999=float(np.quantile(e_hip, 0.999)), median=float(np.median(e_hip)),
                     outliers=n_hip, soft=n_hip, flips=0, n=int(r64.size), scale=float(scale), abs_max=float(np.abs(h - r64).max() / (scale + 1e-30)),
                     allowed=n_ora + allowed_flip, allowed_flips=allowed_flip, outlier_frac_used=n_hip / r64.size, soft_frac_used=0.0, flip_frac_used=0.0,
                     rtol=rtol, soft_max=SOFT_MAX, oracle_over=n_ora, judged="closer")
        assert np.abs(h - r64).max() <= FLIP_ABS_MAX * scale, f"{name}: an entry is off by more than {FLIP_ABS_MAX} x max|ref| from the exact sums"
        if n_hip > n_ora + allowed_flip:
            assert band is not None, f"{name}: over the parity budget ({first}) and NOT closer to the exact sums than the float32 oracle: {n_hip} entries over {rtol} against {n_ora}"
            lo, hi = band()
            st = envelope_residue({"x": np.asarray(hip).ravel()}, {"x": np.asarray(ref32).ravel()}, {"x": np.asarray(lo).ravel()}, {"x": np.asarray(hi).ravel()}, "x", rtol, floor)
            print(f"[parity] {name:18s} vs the reference's band: HIP over: {st['hip_over']}, of those where the oracle moves: {st['hip_over_where_oracle_moves_half']}, "
                  f"band itself over: {st['oracle_band_over']}, worst outside: {st['worst_outside_in_widths']:.2f} widths")
            entry.update(name=name + " (vs the band)", judged="band", oracle_band_over=st["oracle_band_over"], worst_outside_in_widths=st["worst_outside_in_widths"])
            assert st["hip_over_where_oracle_moves_half"] == st["hip_over"] and st["worst_outside_in_widths"] <= max_widths, \
                f"{name}: over the parity budget ({first}), not closer to the exact sums ({n_hip} vs {n_ora}) and not in the reference's band: {st}"
        PARITY_LOG.append(entry)
        return entry


GRAD_KEYS_SR = ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations")


# ---- surfel variant (BASELINE config 5) ---------------------------------------------------------------------------
GRAD_KEYS_SURFEL = ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations")


def surfel_scene(kind, P, H, seed, random_view=True):
    import lidargs_scenes as sc
    s = sc.make_scene(kind, P, H, seed, random_view=random_view)
    s["scales"] = np.ascontiguousarray(s["scales"][:, :2])
    return s


def surfel_upstream_grads(H, W, seed):
    rng = np.random.default_rng(seed + 200)
    return rng.normal(size=(2, H, W)).astype(np.float32), rng.normal(size=(7, H, W)).astype(np.float32)


def hip_surfel_forward_backward(scene, W, H, grads=None, far=80, near=0, scale_modifier=1.0, device="cuda"):
    import torch
    from diff_lidargs_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    st = to_torch(scene, device)
    P = st["means3D"].shape[0]
    leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((P, 4), dtype=torch.float32, device=device, requires_grad=True)
    settings = GaussianRasterizationSettings(
        image_height=int(H), image_width=int(W), bg=st["bg"], scale_modifier=scale_modifier, depth_threshold=0.0,
        viewmatrix=st["viewmatrix"], projmatrix=torch.eye(4, device=device), sh_degree=1, campos=torch.zeros(3, device=device),
        prefiltered=False, beam_inclinations=st["beams"], lidar_far=int(far), lidar_near=int(near), debug=False)
    rast = GaussianRasterizer(settings)
    tm = None
    if scene.get("transMat") is not None:
        # precomputed blend rows next to scales / rotations: only the functional entry point takes both (the module's forward insists
        # on exactly one, as the reference's does, R2/diff_lidargs_surfel_rasterization/__init__.py:236-240)
        from diff_lidargs_surfel_rasterization import rasterize_gaussians
        tm = st["transMat"].reshape(P, 9).clone().requires_grad_(True)
        color, radii, others, pixels = rasterize_gaussians(leaves["means3D"], means2D, torch.empty(0, device=device), leaves["colors"],
                                                           leaves["opacities"], leaves["scales"], leaves["rotations"], tm, settings)
    else:
        color, radii, others, pixels = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=None,
                                            colors_precomp=leaves["colors"], scales=leaves["scales"], rotations=leaves["rotations"])
    out = dict(color=color.detach().cpu().numpy(), others=others.detach().cpu().numpy(), radii=radii.cpu().numpy(),
               pixels=pixels.cpu().numpy(), rasterizer=rast)
    if grads is not None:
        gc, go = (torch.from_numpy(g).to(device) for g in grads)
        torch.autograd.backward([color, others], [gc, go])
        out.update(dL_dmeans3D=leaves["means3D"].grad.cpu().numpy(), dL_dmeans2D=means2D.grad.cpu().numpy(),
                   dL_dcolors=leaves["colors"].grad.cpu().numpy(), dL_dopacity=leaves["opacities"].grad.cpu().numpy(),
                   dL_dscales=leaves["scales"].grad.cpu().numpy(), dL_drotations=leaves["rotations"].grad.cpu().numpy())
        if tm is not None:
            out["dL_dtransMat"] = tm.grad.cpu().numpy()
    return out


def oracle_surfel_forward_backward(scene, W, H, grads=None, far=80, near=0, scale_modifier=1.0):
    from oracle import lgo_surfel
    f = lgo_surfel.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"],
                           scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"], scale_modifier=scale_modifier, far=far, near=near,
                           transMat_precomp=scene.get("transMat"))
    out = dict(color=f.color, others=f.others, radii=f.radii, fwd=f)
    if grads is not None:
        out.update(lgo_surfel.backward(f, *grads))
    return out


# ---- the band of the reference itself, entry by entry (tests/test_sweep_residue_gpu.py, tools/sweep_envelope.py) ---------------------
ULP_MODES = ((1, 1), (1, 2), (2, 0), (3, 0))     # lgo_set_ulp_perturbation: pseudo-random (two seeds), all up, all down


def oracle_envelope(scene, W, H, grads, kw, keys):
    """The plain oracle and, per array of `keys`, the entry-wise [min, max] over it, four runs with every cos / sin / atan2 / tan / exp
    result moved inside its CUDA-libdevice error bound (oracle/lidargs_oracle.c lgo_set_ulp_perturbation; tests/test_ulp_band_cpu.py) and one
    with the backward's pixels visited in reverse order (lgo_set_reverse_pixel_order: the reference's float atomics)."""
    import ctypes as C
    from oracle import lgo
    L = lgo.lib()
    runs = [oracle_forward_backward(scene, W, H, grads, **kw)]
    for mode, sd in ULP_MODES:
        L.lgo_set_ulp_perturbation(C.c_int(mode), C.c_uint(sd))
        try:
            runs.append(oracle_forward_backward(scene, W, H, grads, **kw))
        finally:
            L.lgo_set_ulp_perturbation(C.c_int(0), C.c_uint(0))
    # ... and once with the backward's pixels visited in reverse: the reference adds a Gaussian's per-pixel terms with float atomics in
    # scheduling order (R3/cr/backward.cu:702-788), so the ORDER of those fp32 sums is part of its band too (round 5)
    L.lgo_set_reverse_pixel_order(C.c_int(1))
    try:
        runs.append(oracle_forward_backward(scene, W, H, grads, **kw))
    finally:
        L.lgo_set_reverse_pixel_order(C.c_int(0))
    # ... and once as nvcc compiles the reference by default: a * b + c contracted into one rounding (-fmad=true)
    with lgo.fma_build():
        runs.append(oracle_forward_backward(scene, W, H, grads, **kw))
    lo = {k: np.min([np.asarray(r[k], np.float64) for r in runs], 0) for k in keys}
    hi = {k: np.max([np.asarray(r[k], np.float64) for r in runs], 0) for k in keys}
    return runs[0], lo, hi


def oracle_surfel_envelope(scene, W, H, grads, kw, keys):
    """oracle_envelope for the surfel oracle (lidargs_surfel_oracle.c sfo_set_ulp_perturbation)."""
    import ctypes as C
    from oracle import lgo
    L = lgo.lib()
    runs = [oracle_surfel_forward_backward(scene, W, H, grads, **kw)]
    for mode, sd in ULP_MODES:
        L.sfo_set_ulp_perturbation(C.c_int(mode), C.c_uint(sd))
        try:
            runs.append(oracle_surfel_forward_backward(scene, W, H, grads, **kw))
        finally:
            L.sfo_set_ulp_perturbation(C.c_int(0), C.c_uint(0))
    L.sfo_set_reverse_pixel_order(C.c_int(1))                          # the atomics' summation order (as oracle_envelope)
    try:
        runs.append(oracle_surfel_forward_backward(scene, W, H, grads, **kw))
    finally:
        L.sfo_set_reverse_pixel_order(C.c_int(0))
    with lgo.fma_build():                                              # multiply-adds contracted, as nvcc's default
        runs.append(oracle_surfel_forward_backward(scene, W, H, grads, **kw))
    lo = {k: np.min([np.asarray(r[k], np.float64) for r in runs], 0) for k in keys}
    hi = {k: np.max([np.asarray(r[k], np.float64) for r in runs], 0) for k in keys}
    return runs[0], lo, hi


def envelope_verdict(hip, base, lo, hi, keys, max_widths=1.0):
    """The assertions of tests/test_sweep_residue_gpu.py as a verdict on a whole scene: on every array of `keys`, every entry where HIP is off
    by more than 1e-4 is one the oracle itself moves by more than 0.5e-4 between its seven conforming evaluations (four ulp perturbations of
    cos / sin / atan2 / tan / exp inside their CUDA-libdevice bounds, the reversed summation order of the backward's atomics, multiply-adds
    contracted as nvcc's default -fmad=true does), and HIP lies
    within `max_widths` local widths of their envelope there (seven runs sample the band, they do not bound it: an eighth conforming run
    falls outside the range of seven with probability 1/4).  -> (inside, {key: envelope_residue(...)})."""
    stats = {k: envelope_residue(hip, base, lo, hi, k) for k in keys}
    ok = all(st["hip_over_where_oracle_moves_half"] == st["hip_over"] and st["worst_outside_in_widths"] <= max_widths for st in stats.values())
    return ok, stats


def envelope_residue(hip, base, lo, hi, k, rtol=RTOL, floor=FLOOR):
    """Where HIP is off by more than rtol (parity()'s metric) on array k: does the reference's own band reach that far there, and how
    far outside the envelope [lo, hi] does HIP lie, in units of (the envelope's width at the entry + the rtol bar)?"""
    h = np.asarray(hip[k], np.float64); r = np.asarray(base[k], np.float64)
    den = np.abs(r) + floor * np.abs(r).max() + 1e-30
    off = np.abs(h - r) / den > rtol
    band = (hi[k] - lo[k]) / den
    out = np.maximum(np.maximum(lo[k] - h, h - hi[k]), 0.0)
    rel_out = out / ((hi[k] - lo[k]) + rtol * den)
    return dict(n=int(r.size), hip_over=int(off.sum()), oracle_band_over=int((band > rtol).sum()),
                hip_over_where_oracle_moves=int((off & (band > rtol)).sum()), hip_over_where_oracle_moves_half=int((off & (band > 0.5 * rtol)).sum()),
                worst_outside_in_widths=float(rel_out[off].max()) if off.any() else 0.0, worst_outside_anywhere=float(rel_out.max()))
