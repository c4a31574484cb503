"""GPU parity of the surfel variant (BASELINE config 5): the HIP path, driven through the
drop-in package `diff_lidargs_surfel_rasterization` and the C ABI (lidargs_surfel_*), against the CPU restatement of R2
(oracle/lidargs_surfel_oracle.c).  Tolerance: 1e-4 relative fp32 with the outlier budget of tests/util.py."""
import numpy as np
import pytest

from util import (GRAD_KEYS_SURFEL, hip_surfel_forward_backward, oracle_surfel_forward_backward, parity, surfel_scene,
                  surfel_upstream_grads)

pytestmark = pytest.mark.gpu

OTHERS = ("depth", "alpha", "normal_x", "normal_y", "normal_z", "median_depth", "distortion")
DISTORTION_SOFT_FRAC = 7.5e-4
MEDIAN_TIE_FRAC = 1e-3      # pixels whose T sits within rounding of 0.5 when a surfel is blended: the median-depth selection may pick the neighbour


def _check(scene, W, H, seed, grads=True, **kw):
    g = surfel_upstream_grads(H, W, seed) if grads else None
    if grads:
        # The median depth is a selection (the surfel at which T crosses 0.5).  A pixel whose T lands within rounding of
        # 0.5 selects a neighbouring surfel, which moves a whole upstream-gradient unit from one surfel to another; such
        # near-tie pixels (counted and bounded below) are taken out of the median plane's upstream gradient for both sides.
        h0 = hip_surfel_forward_backward(scene, W, H, None, **kw)
        r0 = oracle_surfel_forward_backward(scene, W, H, None, **kw)
        tie = np.abs(h0["others"][5] - r0["others"][5]) > 1e-4 * (np.abs(r0["others"][5]) + 1e-3)
        n_tie, allowed = int(tie.sum()), max(2, int(MEDIAN_TIE_FRAC * tie.size))
        print(f"[surfel] median-depth near-tie pixels removed from the upstream gradient: {n_tie} of {tie.size} (allowed {allowed})")
        assert n_tie <= allowed
        g[1][5][tie] = 0.0
    hip = hip_surfel_forward_backward(scene, W, H, g, **kw)
    ref = oracle_surfel_forward_backward(scene, W, H, g, **kw)
    nbad = int((hip["radii"] != ref["radii"]).sum())
    assert nbad <= max(1, len(ref["radii"]) // 2000), f"radii differ for {nbad} surfels"
    parity("color", hip["color"], ref["color"])
    for k, name in enumerate(OTHERS):
        if name == "median_depth":
            # the median depth is a selection (the depth of the surfel at which T crosses 0.5): a pixel whose T lands
            # within an ulp of 0.5 picks the neighbour surfel; those pixels are bounded in number, not in size
            d = np.abs(hip["others"][k] - ref["others"][k]) > 1e-4 * (np.abs(ref["others"][k]) + 1e-3)
            assert int(d.sum()) <= max(2, int(MEDIAN_TIE_FRAC * d.size)), f"median depth differs on {int(d.sum())} of {d.size} pixels"
        elif name == "distortion":
            # sum of (m^2 (1-T) + M2 - 2 m M1) w (R2/cr/forward.cu:497-499): a difference of O(1) terms (m in [0,1)) that
            # cancels 4-5 digits, evaluated in fp32 by the reference.  Its rounding error is absolute, ~ulp(1) per blended
            # surfel, so the 1e-4 relative bar applies to the magnitude of the terms, not of the (tiny) difference.
            # The terms' magnitude comes from the oracle's own running sums M1 = sum m w, M2 = sum m^2 w (image state planes 1, 2,
            # R2/cr/rasterizer_impl.cu:177): scale = max(M1^2, M2) over the image (<= 1), not a constant.
            acc = ref["fwd"].array("accum").reshape(3, -1)
            scale = float(max(np.square(acc[1]).max(), acc[2].max(), 1e-6))
            # Its soft class is the suite's widest (round 3: 80 of 169 600 pixels between 1e-4 and 1.6e-4 of that scale, on config 5's
            # full-size frame): stated here rather than in the general budget -- what was used plus 50 %.
            parity("others." + name, hip["others"][k], ref["others"][k], scale=scale, soft_frac=DISTORTION_SOFT_FRAC)
        else:
            parity("others." + name, hip["others"][k], ref["others"][k])
    if grads:
        for k in GRAD_KEYS_SURFEL:
            parity(k, hip[k], ref[k])
    return hip, ref


def test_surfel_shell_small():
    _check(surfel_scene("shell", 3000, 16, 3), 512, 16, 3)


def test_surfel_config1_street():
    _check(surfel_scene("street", 10_000, 16, 1), 512, 16, 1)


def test_surfel_identity_view_and_background():
    sc = surfel_scene("shell", 2000, 16, 5, random_view=False)
    sc["bg"] = np.array([0.25, -0.5], np.float32)
    _check(sc, 512, 16, 5)


def test_surfel_scale_modifier_and_range_window():
    _check(surfel_scene("shell", 3000, 16, 7), 512, 16, 7, scale_modifier=0.7, far=60, near=2)


def test_surfel_ragged_width_and_height():
    # W not a multiple of 16, H not a multiple of the patch height
    _check(surfel_scene("shell", 2500, 18, 9), 500, 18, 9)


def test_surfel_rect_upper_tile_edge_takes_two_roundings():
    """R2/cr/auxiliary.h:107 has the same `p.x + rx + BLOCK_X - 1` as the 3-D rasterizer: two fp32 roundings (see
    test_parity_gpu.py::test_rect_upper_tile_edge_takes_two_roundings).  A surfel centred two ulps under column 16."""
    H, W = 16, 31
    scene = surfel_scene("shell", 4, H, 3, random_view=False)
    scene["means3D"][2, 1] = np.float32(-4.874938011169434)
    hip, ref = _check(scene, W, H, 3)
    assert ref["fwd"].array("means2D").reshape(-1, 2)[2, 0] == np.float32(15.999998092651367)
    assert np.array_equal(hip["radii"], ref["radii"])
    assert ref["others"][1][9, 16] > 0.05 and abs(hip["others"][1][9, 16] - ref["others"][1][9, 16]) < 1e-5   # the one pixel it covers


def test_surfel_config5_shape_crop():
    # config 5 geometry (64 x 2650) with a Gaussian count the oracle finishes in seconds
    _check(surfel_scene("shell", 20_000, 64, 11), 2650, 64, 11)


def _seam_scene(P, H, seed):
    """Surfels on both sides of the panorama's seam (azimuth +-pi: view-space -x axis), identity view: the +-3 sigma end points of those
    next to it project to the other end of the image, the reference rect then covers every tile column (R2/cr/forward.cu:177-215), and
    the pixels that take the surfel are in the first AND the last columns."""
    sc = surfel_scene("shell", P, H, seed, random_view=False)
    rng = np.random.default_rng(seed + 31)
    r = rng.uniform(4.0, 40.0, P)
    az = np.pi + rng.normal(scale=0.3, size=P)
    el = rng.uniform(float(sc["beams"][0]), float(sc["beams"][-1]), P)
    sc["means3D"] = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], 1).astype(np.float32)
    return sc


def test_surfel_seam_of_the_panorama():
    """The binned span of a surfel at the seam runs across it (tile columns gx - a .. gx - 1, 0 .. b: sf_prune + the emit's wrap)."""
    hip, ref = _check(_seam_scene(800, 16, 21), 512, 16, 21)
    assert float(ref["others"][1][:, :8].max()) > 0.05 and float(ref["others"][1][:, -8:].max()) > 0.05     # both ends of the image are covered


@pytest.mark.parametrize("case", ["street", "seam", "thin"])
def test_surfel_pruning_is_invisible(case):
    """LIDARGS_NO_PRUNE=1 bins every tile / row of the reference rect (what the reference does); the default drops those no pixel can
    take (surfel.hip sf_prune).  Same images and gradients either way (up to the summation grouping: the lists are cut into segments by
    length), strictly fewer instances binned."""
    import os, subprocess, sys, tempfile
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
from test_surfel_gpu import _seam_scene
from util import hip_surfel_forward_backward, surfel_scene, surfel_upstream_grads, GRAD_KEYS_SURFEL
from diff_lidargs_rasterization import _C
case = %r
H, W = 64, 2650
scene = _seam_scene(60_000, H, 5) if case == "seam" else surfel_scene("street", 150_000, H, 5, random_view=(case == "thin"))
if case == "thin":
    scene["opacities"] = (scene["opacities"] * np.float32(0.1)).astype(np.float32)
hip = hip_surfel_forward_backward(scene, W, H, surfel_upstream_grads(H, W, 5))
np.savez(sys.argv[1], instances=_C.last_counters()["instances"], **{k: hip[k] for k in ("color", "others", "radii") + GRAD_KEYS_SURFEL})
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (root, os.path.join(root, "lidar-gs_amd"), os.path.join(root, "tests"), case)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, env in (("pruned", {}), ("unpruned", {"LIDARGS_NO_PRUNE": "1"})):
            out = os.path.join(tmp, name + ".npz")
            r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-3000:]
            res[name] = dict(np.load(out))
    a, b = res["pruned"], res["unpruned"]
    print(f"[surfel prune, {case}] instances binned: {int(a['instances'])} pruned, {int(b['instances'])} unpruned")
    assert int(a["instances"]) < 0.8 * int(b["instances"])
    assert np.array_equal(a["radii"], b["radii"])
    for k in ("color", "others") + GRAD_KEYS_SURFEL:
        parity(f"pruned vs unpruned {k}", a[k], b[k], rtol=2e-5, verbose=False)


def test_surfel_outputs_are_written_everywhere():
    """The binding hands the library uninitialised outputs (as the 3-D binding does): every pixel of both images and every row of the
    radii must be written whatever the frame holds.  The caching allocator is primed with NaN / -1 blocks of the outputs' sizes so that
    an element the library skipped shows up: a frame whose surfels cover a corner of the image, and one with everything culled."""
    import torch
    H, W, P = 16, 400, 300
    few = surfel_scene("shell", P, H, 9)
    culled = dict(few); culled["means3D"] = (few["means3D"] * 1000.0).astype(np.float32)
    for scene in (few, culled):
        for shape, dt, val in (((2, H, W), torch.float32, float("nan")), ((7, H, W), torch.float32, float("nan")), ((P,), torch.int32, -1),
                               ((2 * P,), torch.int32, -1)):
            t = torch.full(shape, val, dtype=dt, device="cuda"); del t      # its block is what torch.empty gets next
        hip = hip_surfel_forward_backward(scene, W, H, None)
        ref = oracle_surfel_forward_backward(scene, W, H, None)
        assert np.isfinite(hip["color"]).all() and np.isfinite(hip["others"]).all()
        assert (hip["radii"] >= 0).all() and np.array_equal(hip["radii"], ref["radii"])
        parity("color", hip["color"], ref["color"])


def test_surfel_empty_and_all_culled():
    import torch
    sc = surfel_scene("shell", 64, 16, 2)
    far_away = dict(sc)
    far_away["means3D"] = (sc["means3D"] * 1000.0).astype(np.float32)        # beyond lidar_far
    hip = hip_surfel_forward_backward(far_away, 512, 16, surfel_upstream_grads(16, 512, 2))
    assert (hip["radii"] == 0).all()
    assert np.allclose(hip["color"], sc["bg"].reshape(2, 1, 1)) and (hip["others"] == 0).all()
    for k in GRAD_KEYS_SURFEL:
        assert (hip[k] == 0).all(), k
    empty = {k: (v[:0] if k in ("means3D", "colors", "opacities", "scales", "rotations") else v) for k, v in sc.items()}
    hip = hip_surfel_forward_backward(empty, 512, 16, None)
    assert hip["radii"].shape == (0,) and hip["color"].shape == (2, 16, 512)
    torch.cuda.synchronize()


def test_surfel_visible_filter_and_mark_visible():
    import torch
    from oracle import lgo, lgo_surfel
    sc = surfel_scene("street", 5000, 16, 4)
    hip = hip_surfel_forward_backward(sc, 512, 16, None)
    rast = hip["rasterizer"]
    dev = "cuda"
    m3 = torch.from_numpy(sc["means3D"]).to(dev)
    radii = rast.visible_filter(m3, torch.from_numpy(sc["scales"]).to(dev), torch.from_numpy(sc["rotations"]).to(dev)).cpu().numpy()
    ref = lgo_surfel.visible_filter(sc["means3D"], sc["scales"], sc["rotations"], sc["viewmatrix"], sc["beams"], 512, 16)
    assert int((radii != ref).sum()) <= 2
    vis = rast.markVisible(m3).cpu().numpy()
    assert (vis == lgo.mark_visible(sc["means3D"], sc["viewmatrix"])).all()


def test_surfel_missing_colors_raises():
    import torch
    from diff_lidargs_surfel_rasterization import _C
    sc = surfel_scene("shell", 100, 16, 1)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sc.items()}
    e = torch.empty(0, device="cuda")
    with pytest.raises(RuntimeError, match="precomputed Gaussian colors"):
        _C.rasterize_gaussians(t["bg"], t["means3D"], e, t["opacities"], t["scales"], t["rotations"], 1.0, e, t["viewmatrix"],
                               torch.eye(4).cuda(), t["beams"], 16, 512, e, 1, torch.zeros(3).cuda(), False, 80, 0, False)


def test_surfel_transmat_precomp_drives_the_blend():
    """transMat_precomp next to scales / rotations (the functional entry point): the preprocess -- rect, normal, sort depth, pixel
    centre -- still runs on scales / rotations, the blend (forward and backward) on the precomputed rows
    (R2/cr/rasterizer_impl.cu:332, :408), and the rows' own gradient is dL_dtransMat (R2/__init__.py:74-86)."""
    from oracle import lgo_surfel
    W, H, seed = 512, 16, 11
    scene = surfel_scene("shell", 3000, H, seed)
    plain = lgo_surfel.forward(scene["means3D"], scene["colors"], scene["opacities"], scene["scales"], scene["rotations"],
                               scene["viewmatrix"], scene["beams"], W, H, bg=scene["bg"])
    tm = plain.array("transMat").reshape(-1, 9).copy()
    # the rows the preprocess would have built, unchanged: nothing may move
    same = dict(scene); same["transMat"] = tm
    h_same = hip_surfel_forward_backward(same, W, H, None)
    h_plain = hip_surfel_forward_backward(scene, W, H, None)
    live = h_plain["radii"] > 0
    assert np.array_equal(h_same["radii"], h_plain["radii"])
    # (rows of culled surfels are zero in the oracle's state and never read; the live rows are the preprocess's own up to fp32 rounding)
    parity("color (identity rows)", h_same["color"], h_plain["color"])
    # perturbed rows: axes scaled and tilted, centres moved by a few per cent
    rng = np.random.default_rng(seed)
    tm2 = (tm * (1.0 + 0.05 * rng.normal(size=tm.shape)) + 0.01 * rng.normal(size=tm.shape) * live[:, None]).astype(np.float32)
    pert = dict(scene); pert["transMat"] = tm2
    hip, ref = _check(pert, W, H, seed)
    assert np.array_equal(hip["radii"], h_plain["radii"])               # the rect does not look at the rows
    assert np.abs(hip["color"] - h_plain["color"]).max() > 1e-3         # ... and the image does
    parity("dL_dtransMat", hip["dL_dtransMat"], ref["dL_dtransMat"])


def test_surfel_transmat_precomp_alone_is_refused():
    import torch
    from diff_lidargs_surfel_rasterization import _C
    sc = surfel_scene("shell", 100, 16, 1)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in sc.items()}
    e = torch.empty(0, device="cuda")
    with pytest.raises(RuntimeError, match="without scales and rotations"):
        _C.rasterize_gaussians(t["bg"], t["means3D"], t["colors"], t["opacities"], e, e, 1.0, torch.zeros((100, 9), device="cuda"),
                               t["viewmatrix"], torch.eye(4).cuda(), t["beams"], 16, 512, e, 1, torch.zeros(3).cuda(), False, 80, 0, False)


def test_surfel_config5_fullsize_properties():
    """BASELINE config 5 at full size (2 M surfels, 64 x 2650): size-independent properties of the blend."""
    import torch
    import lidargs_scenes as sc
    kind, P, H, W, seed = sc.BASELINE_CONFIGS["cfg5"]
    scene = surfel_scene(kind, P, H, seed, random_view=False)
    g = surfel_upstream_grads(H, W, seed)
    a = hip_surfel_forward_backward(scene, W, H, g)
    b = hip_surfel_forward_backward(scene, W, H, g)
    alpha, depth, nrm, med, dist = a["others"][1], a["others"][0], a["others"][2:5], a["others"][5], a["others"][6]
    assert np.isfinite(a["color"]).all() and np.isfinite(a["others"]).all()
    assert (alpha >= 0).all() and (alpha <= 1.0).all() and alpha.mean() > 0.5
    assert (np.linalg.norm(nrm, axis=0) <= alpha + 1e-4).all()                    # sum of w_i n_i, unit normals
    assert (depth[alpha > 0] > 0).all() and (depth[alpha == 0] == 0).all()
    assert ((med == 0) | (med >= 0.2)).all() and (np.abs(dist) < 1e-1).all()
    # the forward is a deterministic function of its inputs (no atomics): bit-identical on a second run
    for k in ("color", "others", "radii"):
        assert np.array_equal(a[k], b[k]), k
    # gradients: culled surfels get exact zeros, visible ones finite values; atomics make them order-dependent only in the last bits
    dead = a["radii"] == 0
    for k in GRAD_KEYS_SURFEL:
        assert np.isfinite(a[k]).all(), k
        assert (a[k][dead] == 0).all(), k
        scale = np.abs(a[k]).max()
        assert np.abs(a[k] - b[k]).max() <= 1e-4 * scale, k
    torch.cuda.synchronize()


@pytest.mark.parametrize("opacity_scale", [1.0, 0.1], ids=["cfg5", "cfg5_thin"])
def test_surfel_config5_fullsize_wedge_matches_oracle(opacity_scale):
    """BASELINE config 5 at full size, value by value: the FULL 2 M-surfel frame is rendered on the GPU (the production segment plan:
    96-entry segments, 45 slots, gated first round), the oracle renders every surfel that can reach an azimuth wedge; the wedge's
    pixels and the gradients of the surfels whose whole rect lies inside it are compared under the rules of `_check`.
    `cfg5_thin` (round 6): opacities x 0.1 -- the semi-transparent, early-training regime where no list saturates early."""
    import lidargs_scenes as sc
    kind, P, H, W, seed = sc.BASELINE_CONFIGS["cfg5"]
    scene = surfel_scene(kind, P, H, seed, random_view=False)
    scene["opacities"] = (scene["opacities"] * np.float32(opacity_scale)).astype(np.float32)
    g = surfel_upstream_grads(H, W, seed)
    g[1][5] = 0.0            # the median-depth plane is a selection: its pixels are compared below, its gradient in the small cases
    hip = hip_surfel_forward_backward(scene, W, H, g)
    vm = scene["viewmatrix"].astype(np.float64)
    p = scene["means3D"].astype(np.float64) @ vm[:3, :3] + vm[3, :3]
    pc = (np.pi - np.arctan2(p[:, 1], p[:, 0])) / (2 * np.pi / W)
    reach = np.where(hip["radii"] > 0, hip["radii"].astype(np.float64) + 32.0, 64.0)
    for centre in (W // 2, W // 4 + 8):
        c0 = (centre - 96) // 16 * 16
        c1 = c0 + 192
        keep = (pc >= c0 - reach) & (pc <= c1 + reach)
        sub = dict(scene)
        for k in ("means3D", "scales", "rotations", "opacities", "colors"):
            sub[k] = np.ascontiguousarray(scene[k][keep])
        ref = oracle_surfel_forward_backward(sub, W, H, g)
        rows = np.nonzero(keep)[0]
        mism = int((hip["radii"][rows] != ref["radii"]).sum())
        print(f"[wedge] cfg5 columns [{c0},{c1}): {keep.sum()} surfels in reach, radii mismatches {mism}")
        assert mism <= max(1, int(keep.sum()) // 2000)
        parity(f"cfg5.color[{c0}:{c1}]", hip["color"][..., c0:c1], ref["color"][..., c0:c1])
        for k, name in enumerate(OTHERS):
            a, b = hip["others"][k][:, c0:c1], ref["others"][k][:, c0:c1]
            if name == "median_depth":
                d = np.abs(a - b) > 1e-4 * (np.abs(b) + 1e-3)
                assert d.mean() < 2e-3, f"median depth differs on {d.mean():.2%} of the wedge's pixels"
            elif name == "distortion":
                parity("cfg5.others." + name, a, b, scale=10.0)
            else:
                parity("cfg5.others." + name, a, b)
        m2 = ref["fwd"].array("means2D").reshape(-1, 2)
        rx = ref["fwd"].array("radii_xy").reshape(-1, 2)[:, 0].astype(np.float64)
        x_lo = np.floor((m2[:, 0] - rx) / 16.0) * 16          # R2/cr/auxiliary.h:99-112: tile columns as in R3
        x_hi = np.floor((m2[:, 0] + rx + 15.0) / 16.0) * 16
        inside = (ref["radii"] > 0) & (hip["radii"][rows] > 0) & (x_lo >= c0) & (x_hi <= c1)
        print(f"[wedge] cfg5 columns [{c0},{c1}): {int(inside.sum())} surfels with their whole rect inside")
        assert inside.sum() > 2000
        if opacity_scale != 1.0:
            from util import oracle_backward_exact_sums, parity_or_closer
            ref64 = oracle_backward_exact_sums(ref, g, surfel=True)
            for k in GRAD_KEYS_SURFEL:
                parity_or_closer(f"cfg5_thin.{k}[wedge]", hip[k][rows[inside]], ref[k][inside], ref64[k][inside])
            continue
        for k in GRAD_KEYS_SURFEL:
            parity(f"cfg5.{k}[wedge]", hip[k][rows[inside]], ref[k][inside])
