"""Synthetic LiDAR Gaussian scenes.  numpy only, seeded, deterministic.

No dataset ships with the reference (its .gitignore excludes `data`), so every parity case
and bench workload is generated here.  Conventions follow the reference's call site
gaussian_renderer/__init__.py:145-179: colours [P,2] (intensity, ray-drop), opacities [P,1],
scales [P,3], unit quaternions [P,4] as (r,x,y,z), viewmatrix = TRANSPOSED world->lidar 4x4
(scene/cameras.py:56), beam_inclinations ascending radians, lidar_far=80, lidar_near=0.
"""
import numpy as np

BASELINE_CONFIGS = {
    # name: (scene kind, P, H, W, seed)  -- BASELINE.json `configs`, SURVEY.md section 8 sizes
    "cfg1": ("shell", 10_000, 16, 512, 1),
    "cfg2": ("street", 500_000, 64, 2650, 2),
    "cfg3": ("street", 2_000_000, 64, 2650, 3),
    "cfg4": ("shell", 8_000_000, 128, 4096, 4),
    "cfg5": ("street", 2_000_000, 64, 2650, 5),     # surfel variant: scales[:, :2] are the two surfel axes
    # not a BASELINE.json config: config 4's size with the street scene's statistics (long lists, early saturation), used to
    # check that plan / tile-height choices tuned on cfg4's isotropic shell do not hurt the other kind of big frame
    "cfg4_street": ("street", 8_000_000, 128, 4096, 6),
}


def beam_inclinations(H, lo_deg=-17.6, hi_deg=2.4):
    """Waymo-top-like vertical FOV, ascending radians (R3/cr/forward.cu:337 needs ascending)."""
    return np.deg2rad(np.linspace(lo_deg, hi_deg, H)).astype(np.float32)


BEAM_TABLES = ("uniform", "waymo", "neartie")


def beam_table(H, kind="uniform", lo_deg=-17.6, hi_deg=2.4):
    """Beam-inclination tables, ascending float32 radians.

    uniform  np.linspace over the FOV (also what the reference's get_beam_inclinations gives, utils/lidar_utils.py:296-299).
    waymo    what the Waymo configs really feed the kernel: a MEASURED, non-uniform table read from the dataset json
             (scene/dataset_readers.py:358-359).  No dataset ships, so this is a stand-in with the real table's properties: same FOV,
             gaps shrinking smoothly from the bottom beam to the top one by 4x (the top-lidar is densest near the horizon), and a
             deterministic +-15 % wobble of every gap on top (a measured table is not smooth).  K1 bisects the table
             (R3/cr/auxiliary.h:41-63) and sizes the row radius from the LOCAL gap (R3/cr/forward.cu:361), so unequal gaps are the
             input property that uniform tables never exercise.
    neartie  the uniform table with two neighbouring beams 2e-5 rad apart (a degenerate calibration): the local gap's tangent is
             tiny, the row radius of every Gaussian landing in that interval explodes to the whole image height.
    """
    if kind == "uniform" or H < 3:
        return beam_inclinations(H, lo_deg, hi_deg)
    if kind == "waymo":
        w = 0.6                                                     # gap(bottom) / gap(top) = (1 + w) / (1 - w) = 4
        t = (np.arange(H - 1, dtype=np.float64) + 0.5) / (H - 1)
        gaps = (1.0 - w) + 2.0 * w * (1.0 - t)
        gaps *= 1.0 + 0.15 * np.sin(12.9898 * np.arange(1, H) + 78.233 * H)      # deterministic, no RNG state involved
        f = np.concatenate([[0.0], np.cumsum(gaps)]) / gaps.sum()
        b = np.deg2rad(lo_deg + (hi_deg - lo_deg) * f).astype(np.float32)
    elif kind == "neartie":
        b = beam_inclinations(H, lo_deg, hi_deg).copy()
        k = max(1, (2 * H) // 3)
        b[k] = b[k - 1] + np.float32(2e-5)
    else:
        raise ValueError(f"unknown beam table {kind!r} (one of {BEAM_TABLES})")
    assert np.all(np.diff(b) > 0), "beam table must be strictly ascending"
    return b


def _beams_arg(H, beams):
    if beams is None:
        return beam_inclinations(H)
    if isinstance(beams, str):
        return beam_table(H, beams)
    b = np.ascontiguousarray(beams, dtype=np.float32)
    assert b.shape == (H,)
    return b


def _rand_quat(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def _quat_mul(a, b):
    ar, ax, ay, az = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    br, bx, by, bz = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([ar * br - ax * bx - ay * by - az * bz,
                     ar * bx + ax * br + ay * bz - az * by,
                     ar * by - ax * bz + ay * br + az * bx,
                     ar * bz + ax * by - ay * bx + az * br], axis=1)


def rigid_viewmatrix(rng=None, max_angle=0.3, max_shift=1.0):
    """[4,4] float32 transposed world->lidar matrix; identity when rng is None."""
    V = np.eye(4, dtype=np.float64)
    if rng is not None:
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        ang = rng.uniform(-max_angle, max_angle)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        t = rng.uniform(-max_shift, max_shift, size=3)
        V[:3, :3] = R.T       # row-vector convention: p_view = [p,1] @ V
        V[3, :3] = t
    return V.astype(np.float32)


def shell_scene(P, H, seed, beams=None):
    """Isotropic 'shell': r~U(5,60), azimuth U(-pi,pi), elevation inside the beam fan."""
    rng = np.random.default_rng(seed)
    beams = _beams_arg(H, beams)
    r = rng.uniform(5.0, 60.0, P)
    az = rng.uniform(-np.pi, np.pi, P)
    el = rng.uniform(float(beams[0]), float(beams[-1]), P)
    xyz = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], axis=1)
    scales = 0.1 * np.exp(0.5 * rng.normal(size=(P, 3)))
    rots = _rand_quat(rng, P)
    return _pack(rng, xyz, scales, rots, beams)


def street_scene(P, H, seed, beams=None):
    """Waymo-static stand-in: 70% ground z=-2 (r<75), 25% two walls y=+-12, 5% clutter;
    flat surfels (0.15,0.15,0.02)*lognormal(0.4) aligned with the surface normal."""
    rng = np.random.default_rng(seed)
    beams = _beams_arg(H, beams)
    n_g = int(0.70 * P); n_w = int(0.25 * P); n_c = P - n_g - n_w
    # ground: uniform in area inside a 75 m disc
    rg = 75.0 * np.sqrt(rng.uniform(0.0, 1.0, n_g)); ag = rng.uniform(-np.pi, np.pi, n_g)
    ground = np.stack([rg * np.cos(ag), rg * np.sin(ag), np.full(n_g, -2.0)], axis=1)
    qg = np.stack([np.cos(ag * 0.5), np.zeros(n_g), np.zeros(n_g), np.sin(ag * 0.5)], axis=1)  # yaw only
    # walls: y = +-12, x in [-75,75], z in [-2,6]; local z -> world y (rotate 90deg about x), random spin
    side = rng.choice([-1.0, 1.0], n_w)
    walls = np.stack([rng.uniform(-75.0, 75.0, n_w), 12.0 * side, rng.uniform(-2.0, 6.0, n_w)], axis=1)
    spin = rng.uniform(-np.pi, np.pi, n_w)
    q_spin = np.stack([np.cos(spin * 0.5), np.zeros(n_w), np.zeros(n_w), np.sin(spin * 0.5)], axis=1)
    q_tilt = np.tile(np.array([[np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]]), (n_w, 1))
    qw = _quat_mul(q_tilt, q_spin)
    # clutter: isotropic blobs
    rc = rng.uniform(5.0, 60.0, n_c); ac = rng.uniform(-np.pi, np.pi, n_c)
    ec = rng.uniform(float(beams[0]), float(beams[-1]), n_c)
    clutter = np.stack([rc * np.cos(ec) * np.cos(ac), rc * np.cos(ec) * np.sin(ac), rc * np.sin(ec)], axis=1)
    qc = _rand_quat(rng, n_c)
    xyz = np.concatenate([ground, walls, clutter], axis=0)
    rots = np.concatenate([qg, qw, qc], axis=0)
    scales = np.array([[0.15, 0.15, 0.02]]) * np.exp(0.4 * rng.normal(size=(P, 3)))
    perm = rng.permutation(P)
    return _pack(rng, xyz[perm], scales[perm], rots[perm], beams)


def _pack(rng, xyz, scales, rots, beams):
    P = xyz.shape[0]
    return dict(
        means3D=xyz.astype(np.float32),
        scales=scales.astype(np.float32),
        rotations=rots.astype(np.float32),
        opacities=rng.uniform(0.1, 1.0, (P, 1)).astype(np.float32),
        colors=rng.uniform(0.0, 1.0, (P, 2)).astype(np.float32),
        beams=beams,
        bg=np.zeros(2, np.float32),
        viewmatrix=rigid_viewmatrix(None),
    )


def make_scene(kind, P, H, seed, random_view=False, beams=None, opacity_scale=1.0):
    """beams: None / "uniform" (SURVEY 8d), "waymo", "neartie" (beam_table) or an explicit ascending float32[H] table.
    opacity_scale: the opacities U(0.1, 1) of SURVEY 8d times this factor.  1 = frames that saturate within a few entries (a trained
    scene); 0.1 / 0.03 = the semi-transparent frames training STARTS in (the reference's opacities are a tanh-MLP output masked > 0,
    gaussian_renderer/__init__.py:60-70): no pixel reaches T < 1e-4 early, every list is walked to its end."""
    s = shell_scene(P, H, seed, beams) if kind == "shell" else street_scene(P, H, seed, beams)
    if random_view:
        s["viewmatrix"] = rigid_viewmatrix(np.random.default_rng(seed + 7))
    if opacity_scale != 1.0:
        s["opacities"] = (s["opacities"] * np.float32(opacity_scale)).astype(np.float32)
    return s


def upstream_grads(H, W, seed):
    """N(0,1) upstream gradients for (color[2,H,W], depth[1,H,W], occ[1,H,W]); seed+100 rule."""
    rng = np.random.default_rng(seed + 100)
    return (rng.normal(size=(2, H, W)).astype(np.float32),
            rng.normal(size=(1, H, W)).astype(np.float32),
            rng.normal(size=(1, H, W)).astype(np.float32))


# ---- anchor models for the decode (generate_neural_gaussians) workloads ---------------------------------------------------
ANCHOR_MLPS = ("opacity", "cov", "color", "raydrop")


def make_anchor_model(N, k, seed, flags=(True, True, True)):
    """Random Scaffold-style anchor model in the reference's default configuration (feat 32, hidden 32, k offsets, 2 colour
    channels): dict of numpy arrays (anchor_feat, anchor, offset, scaling = exp-activated, {mlp}_W1/_b1/_W2/_b2, add_*_dist),
    a camera centre, a visible-anchor mask and the generator (for further draws)."""
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.normal(size=s).astype(np.float32)
    p = dict(anchor_feat=0.5 * f(N, 32), anchor=10.0 * f(N, 3), offset=0.3 * f(N, k, 3), scaling=np.exp(0.3 * f(N, 6) - 1.0).astype(np.float32),
             add_opacity_dist=flags[0], add_cov_dist=flags[1], add_color_dist=flags[2])
    dins = dict(opacity=35 + flags[0], cov=35 + flags[1], color=35 + flags[2], raydrop=35 + flags[2])
    douts = dict(opacity=k, cov=7 * k, color=k, raydrop=k)
    for m in ANCHOR_MLPS:
        p[m + "_W1"], p[m + "_b1"] = f(32, dins[m]) / np.float32(6.0), 0.1 * f(32)
        p[m + "_W2"], p[m + "_b2"] = f(douts[m], 32) / np.float32(5.6), 0.1 * f(douts[m])
    return p, np.array([0.5, -1.0, 2.0], np.float32), rng.random(N) > 0.3, rng


def anchor_model_to_torch(p, device="cuda"):
    """The object generate_neural_gaussians reads (`pc`): tensors with requires_grad and the four nn.Sequential MLPs declared as
    GaussianModel.__init__ declares them (scene/gaussian_model.py:113-142)."""
    import types
    import torch
    from torch import nn
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    pc = types.SimpleNamespace()
    k = p["offset"].shape[1]
    pc.use_feat_bank, pc.appearance_dim, pc.n_offsets, pc.color_channel = False, 0, k, 2
    pc.add_opacity_dist, pc.add_cov_dist, pc.add_color_dist = p["add_opacity_dist"], p["add_cov_dist"], p["add_color_dist"]
    for name, act in (("opacity", nn.Tanh()), ("cov", None), ("color", nn.Sigmoid()), ("raydrop", nn.Sigmoid())):
        W1, W2 = p[name + "_W1"], p[name + "_W2"]
        seq = nn.Sequential(nn.Linear(W1.shape[1], 32), nn.ReLU(True), nn.Linear(32, W2.shape[0]), *([act] if act else [])).to(device)
        with torch.no_grad():
            seq[0].weight.copy_(t(W1)); seq[0].bias.copy_(t(p[name + "_b1"])); seq[2].weight.copy_(t(W2)); seq[2].bias.copy_(t(p[name + "_b2"]))
        setattr(pc, "mlp_" + name, seq); setattr(pc, f"get_{name}_mlp", seq)
    pc._anchor_feat = t(p["anchor_feat"]).requires_grad_(True)
    pc._anchor = t(p["anchor"]).requires_grad_(True); pc.get_anchor = pc._anchor
    pc._offset = t(p["offset"]).requires_grad_(True)
    pc.get_scaling = t(p["scaling"]).requires_grad_(True)
    return pc


def raster_settings(scene_t, W, H, far=80, near=0, scale_modifier=1.0, debug=False):
    """GaussianRasterizationSettings for a scene dict of device tensors, built as gaussian_renderer/__init__.py:150-166 does."""
    import torch
    from diff_lidargs_rasterization import GaussianRasterizationSettings
    dev = scene_t["viewmatrix"].device
    return GaussianRasterizationSettings(
        image_height=int(H), image_width=int(W), tanfovx=1.0, tanfovy=1.0, bg=scene_t["bg"], scale_modifier=scale_modifier,
        viewmatrix=scene_t["viewmatrix"], projmatrix=torch.eye(4, device=dev), sh_degree=1,
        campos=torch.zeros(3, device=dev), prefiltered=False, beam_inclinations=scene_t["beams"], debug=debug,
        lidar_far=int(far), lidar_near=int(near))


def sweep_case(seed, mid=None):
    """The 3-D scene tools/parity_sweep.py draws for `seed` (its surfel scenes, seed % 4 == 3 with H >= 4, are drawn differently and are
    not reproduced here): (scene, W, H, upstream grads, keyword arguments of the rasterizer, description).  One recipe for the sweep's
    repro tool (tools/repro_sweep_seed.py) and for the regression tests built from its residue (tests/test_sweep_residue_gpu.py)."""
    if mid is None:
        mid = seed >= 100000 and seed < 900000
    rng = np.random.default_rng(seed)
    if mid:
        H = int(rng.choice([16, 32, 64])); W = int(rng.choice([900, 1800, 2650])); P = int(rng.integers(20000, 60000))
    else:
        H = int(rng.choice([2, 3, 5, 16, 17, 32, 40, 64])); W = int(rng.integers(1, 700)); P = int(rng.integers(1, 6000))
    if not mid and seed % 7 == 5:
        W = int(rng.integers(4100, 4300)); H = int(rng.choice([2, 3, 16])); P = int(rng.integers(1, 3000))
    if not mid and seed % 11 == 7:
        H = int(rng.choice([130, 272])); W = int(rng.integers(1, 200)); P = int(rng.integers(1, 3000))
    if not mid and seed % 17 == 4:
        H = int(rng.choice([1025, 1100])); W = int(rng.integers(64, 200)); P = int(rng.integers(1, 3000))
    if not mid and seed % 19 == 6:
        H = 1100; W = 4800; P = int(rng.integers(1, 2000))
    kind = "shell" if rng.random() < 0.5 else "street"
    beams = str(rng.choice(["uniform", "waymo", "neartie"])) if H >= 4 else "uniform"
    kw = dict(far=int(rng.choice([80, 30])), near=int(rng.choice([0, 2])), scale_modifier=float(rng.choice([1.0, 0.5, 2.5])))
    if not mid and seed % 13 == 3:
        kw["scale_modifier"] = float(rng.choice([6.0, 12.0, 30.0]))
    scene = make_scene(kind, P, H, seed % 1000, random_view=bool(rng.integers(0, 2)), beams=beams)
    if seed % 23 == 8:
        scene["opacities"] = (scene["opacities"] * np.float32(0.008)).astype(np.float32)
    grads = upstream_grads(H, W, seed % 1000)
    return scene, W, H, grads, kw, dict(seed=seed, kind=kind, P=P, H=H, W=W, beams=beams, **kw)


def sweep_case_any(seed, mid=None):
    """sweep_case for EVERY scene tools/parity_sweep.py draws, its surfel scenes (seed % 4 == 3, H >= 4) and precomputed-covariance
    scenes (seed % 5 == 1) included, with the sweep's own order of random draws: dict(scene, W, H, grads, kw, surfel, cov, desc).
    Surfel scenes: scales are [P,2], grads = (dL_dcolor [2,H,W], dL_dothers [7,H,W]) with the median-depth plane's gradient zeroed."""
    if mid is None:
        mid = seed >= 100000 and seed < 900000
    rng = np.random.default_rng(seed)
    if mid:
        H = int(rng.choice([16, 32, 64])); W = int(rng.choice([900, 1800, 2650])); P = int(rng.integers(20000, 60000))
    else:
        H = int(rng.choice([2, 3, 5, 16, 17, 32, 40, 64])); W = int(rng.integers(1, 700)); P = int(rng.integers(1, 6000))
    if not mid and seed % 7 == 5:
        W = int(rng.integers(4100, 4300)); H = int(rng.choice([2, 3, 16])); P = int(rng.integers(1, 3000))
    if not mid and seed % 11 == 7:
        H = int(rng.choice([130, 272])); W = int(rng.integers(1, 200)); P = int(rng.integers(1, 3000))
    if not mid and seed % 17 == 4:
        H = int(rng.choice([1025, 1100])); W = int(rng.integers(64, 200)); P = int(rng.integers(1, 3000))
    if not mid and seed % 19 == 6:
        H = 1100; W = 4800; P = int(rng.integers(1, 2000))
    kind = "shell" if rng.random() < 0.5 else "street"
    beams = str(rng.choice(["uniform", "waymo", "neartie"])) if H >= 4 else "uniform"
    surfel = (seed % 4 == 3) and H >= 4
    kw = dict(far=int(rng.choice([80, 30])), near=int(rng.choice([0, 2])), scale_modifier=float(rng.choice([1.0, 0.5, 2.5])))
    if not mid and seed % 13 == 3:
        kw["scale_modifier"] = float(rng.choice([6.0, 12.0, 30.0]))
    desc = dict(seed=seed, kind=kind, P=P, H=H, W=W, beams=beams, variant="surfel" if surfel else "3d", **kw)
    cov = None
    if surfel:
        scene = make_scene(kind, P, H, seed % 1000, random_view=bool(rng.integers(0, 2)))
        scene["scales"] = np.ascontiguousarray(scene["scales"][:, :2])
        scene["beams"] = beam_table(H, beams)
        g = np.random.default_rng(seed % 1000 + 200)
        grads = (g.normal(size=(2, H, W)).astype(np.float32), g.normal(size=(7, H, W)).astype(np.float32))
        grads[1][5] = 0.0
    else:
        scene = make_scene(kind, P, H, seed % 1000, random_view=bool(rng.integers(0, 2)), beams=beams)
        if seed % 23 == 8:
            scene["opacities"] = (scene["opacities"] * np.float32(0.008)).astype(np.float32)
        grads = upstream_grads(H, W, seed % 1000)
        if seed % 5 == 1:
            A = rng.normal(size=(P, 3, 3)) * float(scene["scales"].mean())
            S = A @ np.transpose(A, (0, 2, 1)) + 1e-4 * np.eye(3)
            cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1).astype(np.float32)
            desc["cov3D_precomp"] = True
    return dict(scene=scene, W=W, H=H, grads=grads, kw=kw, surfel=surfel, cov=cov, desc=desc)


def anchor_scene(N, k, seed, voxel=0.01, feat_dim=32):
    """Inputs of anchor growing (tests/test_anchor_growing.py, bench.py --workload anchor_growing): N street-like
    anchors on the voxel grid (the reference's initialisation leaves them there, scene/gaussian_model.py:274), k offsets each of up to a
    few dozen voxels, log-space scalings, features, and accumulated gradient norms drawn so that the three levels' thresholds
    (0.0005, 0.001, 0.002) select ~3 %, ~0.35 % and ~0.003 % of the offsets."""
    rng = np.random.default_rng(seed)
    M = N + N // 8
    cells = np.unique(np.round(np.stack([rng.uniform(-60, 60, M), rng.uniform(-12, 12, M), rng.normal(-1.5, 0.4, M)], 1) / voxel), axis=0)
    anchor = (cells[rng.permutation(cells.shape[0])[:N]] * voxel).astype(np.float32)
    N = anchor.shape[0]
    return dict(N=N, k=k, voxel=voxel, anchor=anchor, offset=rng.uniform(-1, 1, (N, k, 3)).astype(np.float32),
                scaling=np.log(rng.uniform(0.02, 0.4, (N, 6))).astype(np.float32), feat=rng.normal(size=(N, feat_dim)).astype(np.float32),
                grads=rng.exponential(0.0002, N * k).astype(np.float32), offset_mask=rng.random(N * k) > 0.3)
