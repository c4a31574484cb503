"""GPU parity tests: the HIP path (through the drop-in package and the C ABI) against the CPU oracle
on identical seeded inputs.  Sizes are chosen so that the oracle finishes in seconds.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import numpy as np
import pytest

import lidargs_scenes as sc
from util import GRAD_KEYS_SR, hip_forward_backward, oracle_forward_backward, parity

pytestmark = pytest.mark.gpu


def _check_radii(hip, ref):
    """Integer outputs: bit-exact, except for Gaussians within an ulp of a ceil()/round() boundary
    (device atan2f/tanf vs host libm); those are counted and must stay below 1e-4 of P (min 1)."""
    mism = int((hip != ref).sum())
    allowed = max(1, int(1e-4 * ref.size))
    print(f"[parity] radii mismatches: {mism} of {ref.size} (allowed {allowed})")
    assert mism <= allowed
    assert ((hip > 0) != (ref > 0)).sum() <= allowed


def _compare_all(hip, ref, keys):
    _check_radii(hip["radii"], ref["radii"])
    parity("color", hip["color"], ref["color"])
    parity("depth", hip["depth"], ref["depth"])
    parity("occ", hip["occ"], ref["occ"])
    for k in keys:
        parity(k, hip[k], ref[k])


CASES = [
    # name, kind, P, H, W, seed, random_view, bg
    ("cfg1_shell", "shell", 10_000, 16, 512, 1, False, (0.0, 0.0)),
    ("cfg1_view", "shell", 10_000, 16, 512, 1, True, (0.0, 0.0)),
    ("street_bg", "street", 20_000, 16, 512, 2, True, (0.3, 0.7)),
    ("ragged", "shell", 6_000, 18, 500, 3, True, (0.1, 0.0)),      # W % 16 != 0, H % 4 != 0
    ("dense64", "street", 60_000, 64, 1000, 4, False, (0.0, 0.0)),  # long per-tile lists, early-out active
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_forward_backward_matches_oracle(case, hip_lib_built):
    name, kind, P, H, W, seed, rv, bg = case
    scene = sc.make_scene(kind, P, H, seed, random_view=rv)
    scene["bg"] = np.array(bg, np.float32)
    grads = sc.upstream_grads(H, W, seed)
    ref = oracle_forward_backward(scene, W, H, grads)
    hip = hip_forward_backward(scene, W, H, grads)
    _compare_all(hip, ref, GRAD_KEYS_SR)


RANGE_SORT_CASES = ["duplicate_ranges", "crowded_bucket", "one_bucket", "two_ranges_only", "mostly_culled"]


@pytest.mark.parametrize("bucket_bits", [10, 11], ids=["1024_intervals", "2048_intervals"])
@pytest.mark.parametrize("case", RANGE_SORT_CASES)
def test_bucketed_range_sort_on_inputs_built_against_it(case, bucket_bits, hip_lib_built, monkeypatch):
    """The range sort of frames above 4096 Gaussians is one bucket pass on the LINEAR range + one launch that finishes every bucket in LDS
    (binning.hip, round 5).  The per-tile order it must produce is (range bits, index) -- R3/cr/rasterizer_impl.cu:103-106, :317-322 -- and
    every pixel of the image depends on it.  Inputs built against it: thousands of Gaussians at bit-identical ranges (the stable order
    among equal keys is the index order), a bucket far over the LDS path's 7168 pairs (20 000 Gaussians inside 2 cm of range beside two
    outliers that stretch the span: sorted through global memory by one workgroup), every Gaussian in ONE bucket at one identical range,
    a span with only two distinct keys, and a frame whose Gaussians are mostly culled (the last bucket: the launch's tail of slices).
    Round 6: frames above 4 M Gaussians take the same path with 2048 intervals (cfg4: 8 M); the oracle cannot run at that size, so the form
    is forced here (LIDARGS_RANGE_SORT_BUCKET_BITS=11) on the same adversarial inputs, and tests/test_fullsize_gpu.py checks cfg4's wedge."""
    monkeypatch.setenv("LIDARGS_RANGE_SORT_BUCKET_BITS", str(bucket_bits))
    H, W, seed = 16, 512, 71
    rng = np.random.default_rng(seed)
    scene = sc.make_scene("shell", 30_000 if case != "crowded_bucket" else 24_000, H, seed, random_view=False)
    m = scene["means3D"].astype(np.float64)
    r = np.linalg.norm(m, axis=1, keepdims=True)
    if case == "duplicate_ranges":
        # ranges snapped to 40 values: points in the x-y plane direction scaled so that |p| (fp32, as the kernel evaluates it) repeats
        target = np.round(r / 1.5) * 1.5 + 3.0
        m = m / r * target
    elif case == "crowded_bucket":
        target = 20.0 + rng.uniform(0.0, 0.02, size=r.shape)
        m = m / r * target
        m[0] = m[0] / np.linalg.norm(m[0]) * 2.5; m[1] = m[1] / np.linalg.norm(m[1]) * 78.0
    elif case == "one_bucket":
        m = m / r * 17.0
    elif case == "two_ranges_only":
        m = m / r * np.where(rng.random(r.shape) < 0.5, 9.0, 31.0)
    elif case == "mostly_culled":
        m = m / r * np.where(rng.random(r.shape) < 0.8, 200.0, r)          # 80 % behind lidar_far
    scene["means3D"] = m.astype(np.float32)
    scene["opacities"] = (scene["opacities"] * np.float32(0.25)).astype(np.float32)     # deep lists stay unsaturated: the order shows in every pixel
    grads = sc.upstream_grads(H, W, seed)
    ref = oracle_forward_backward(scene, W, H, grads)
    hip = hip_forward_backward(scene, W, H, grads)
    _compare_all(hip, ref, GRAD_KEYS_SR)


def test_cov3d_precomp_path(hip_lib_built):
    """cov3D_precomp instead of scales/rotations (R3/cr/forward.cu:307-310, backward :520)."""
    P, H, W, seed = 5_000, 16, 512, 5
    scene = sc.make_scene("shell", P, H, seed, random_view=True)
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(P, 3, 3)) * 0.1
    S = A @ np.transpose(A, (0, 2, 1)) + 1e-3 * np.eye(3)
    cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], axis=1).astype(np.float32)
    grads = sc.upstream_grads(H, W, seed)
    ref = oracle_forward_backward(scene, W, H, grads, cov3D_precomp=cov)
    hip = hip_forward_backward(scene, W, H, grads, cov3D_precomp=cov)
    _compare_all(hip, ref, ("dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dcov3D"))


def test_scale_modifier_and_range_cull(hip_lib_built):
    P, H, W, seed = 8_000, 16, 512, 6
    scene = sc.make_scene("shell", P, H, seed)
    grads = sc.upstream_grads(H, W, seed)
    ref = oracle_forward_backward(scene, W, H, grads, far=40, near=10, scale_modifier=1.7)
    hip = hip_forward_backward(scene, W, H, grads, far=40, near=10, scale_modifier=1.7)
    assert (ref["radii"] == 0).sum() > 0.3 * P      # the cull is exercised
    _compare_all(hip, ref, GRAD_KEYS_SR)


def test_visible_filter_and_mark_visible(hip_lib_built):
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from oracle import lgo
    from util import make_settings, to_torch
    P, H, W, seed = 20_000, 64, 2650, 7
    scene = sc.make_scene("street", P, H, seed, random_view=True)
    st = to_torch(scene)
    rast = GaussianRasterizer(make_settings(st, W, H))
    radii = rast.visible_filter(means3D=st["means3D"], scales=st["scales"], rotations=st["rotations"]).cpu().numpy()
    ref = lgo.visible_filter(scene["means3D"], scene["scales"], scene["rotations"], scene["viewmatrix"], scene["beams"], W, H)
    _check_radii(radii, ref)
    vis = rast.markVisible(st["means3D"]).cpu().numpy()
    assert vis.dtype == np.bool_
    np.testing.assert_array_equal(vis, lgo.mark_visible(scene["means3D"], scene["viewmatrix"]))


def test_empty_and_all_culled(hip_lib_built):
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import make_settings, to_torch
    H, W = 16, 512
    scene = sc.make_scene("shell", 100, H, 8)
    scene["bg"] = np.array([0.25, 0.5], np.float32)
    st = to_torch(scene)
    rast = GaussianRasterizer(make_settings(st, W, H))
    # P == 0: outputs are all zeros, even with a background (R3/rasterize_points.cu:87)
    e = lambda *s: torch.zeros(s, device="cuda")
    c, d, o, r = rast(means3D=e(0, 3), means2D=e(0, 4), opacities=e(0, 1), colors_precomp=e(0, 2), scales=e(0, 3), rotations=e(0, 4))
    assert c.shape == (2, H, W) and float(c.abs().max()) == 0.0 and r.numel() == 0
    # everything beyond lidar_far: background only
    far_scene = dict(st)
    far_scene["means3D"] = st["means3D"] * 100.0
    c, d, o, r = rast(means3D=far_scene["means3D"], means2D=e(100, 4), opacities=st["opacities"], colors_precomp=st["colors"],
                      scales=st["scales"], rotations=st["rotations"])
    assert int((r > 0).sum()) == 0
    np.testing.assert_allclose(c[0].cpu().numpy(), 0.25)
    np.testing.assert_allclose(c[1].cpu().numpy(), 0.5)
    assert float(d.abs().max()) == 0.0 and float(o.abs().max()) == 0.0


def test_argument_errors(hip_lib_built):
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import make_settings, to_torch
    scene = sc.make_scene("shell", 10, 16, 9)
    st = to_torch(scene)
    rast = GaussianRasterizer(make_settings(st, 512, 16))
    m2 = torch.zeros(10, 4, device="cuda")
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(means3D=st["means3D"], means2D=m2, opacities=st["opacities"], scales=st["scales"], rotations=st["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=st["means3D"], means2D=m2, opacities=st["opacities"], colors_precomp=st["colors"])
    with pytest.raises(RuntimeError, match="num_points, 3"):
        rast(means3D=st["means3D"][:, :2], means2D=m2, opacities=st["opacities"], colors_precomp=st["colors"],
             scales=st["scales"], rotations=st["rotations"])
    with pytest.raises(RuntimeError, match="precomputed Gaussian colors"):
        # SHs without precomputed colours: the LiDAR build has NUM_CHANNELS != 3 (R3/cr/rasterizer_impl.cu:249-252)
        rast(means3D=st["means3D"], means2D=m2, opacities=st["opacities"], shs=torch.zeros(10, 4, 3, device="cuda"),
             scales=st["scales"], rotations=st["rotations"])


@pytest.mark.parametrize("tile_rows", [8, 16, 32])
def test_tile_height_variants(tile_rows, hip_lib_built):
    """The list-tile height (LIDARGS_TILE_ROWS) is an internal choice: results must not depend on it."""
    import os, subprocess, sys, json, tempfile
    code = r"""
import sys, json, numpy as np
sys.path[:0] = [%r, %r, %r]
import lidargs_scenes as sc
from util import hip_forward_backward, oracle_forward_backward, parity, GRAD_KEYS_SR
scene = sc.make_scene("street", 30000, 32, 10, random_view=True)
grads = sc.upstream_grads(32, 800, 10)
ref = oracle_forward_backward(scene, 800, 32, grads)
hip = hip_forward_backward(scene, 800, 32, grads)
for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
    parity(k, hip[k], ref[k])
print("OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (root, os.path.join(root, "lidar-gs_amd"), os.path.join(root, "tests"))
    env = dict(os.environ, LIDARGS_TILE_ROWS=str(tile_rows))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "OK" in r.stdout


@pytest.mark.parametrize("env", [{"LIDARGS_HEAD": "1"}, {"LIDARGS_HEAD": "1", "LIDARGS_ROUNDS": "2,6"}, {"LIDARGS_HEAD": "0", "LIDARGS_SEG_LEN": "128"},
                                 {"LIDARGS_P2_GROUP": "3"}, {"LIDARGS_SORT_ITEMS": "16"}, {"LIDARGS_RANGE_SORT_BITS": "11"},
                                 {"LIDARGS_FUSED": "0"}, {"LIDARGS_FUSED_WAVES": "4"}, {"LIDARGS_FUSED_WAVES": "16"},
                                 {"LIDARGS_FUSED": "1", "LIDARGS_SEG_LEN": "128", "LIDARGS_MAX_SEGMENTS": "33"}, {"LIDARGS_FUSED": "0", "LIDARGS_HEAD": "1"},
                                 {"LIDARGS_TILE_KEY32": "1"}, {"LIDARGS_NO_SMALL_SORT": "1"}, {"LIDARGS_RANGE_SORT_FULL": "1"},
                                 {"LIDARGS_FUSED": "0", "LIDARGS_WORK_LISTS": "0"}, {"LIDARGS_SMALL_SORT_MAX": "16384"}, {"LIDARGS_RANGE_SORT_BUCKETS": "0"},
                                 {"LIDARGS_RANGE_SORT_BUCKETS": "0", "LIDARGS_RANGE_SORT_FULL": "1"}, {}],
                         ids=["head5", "head2_rounds26", "nohead_seg128", "pass2_groups_of_3", "sort_blocks_4096", "sort_digits_11",
                              "five_launch_forward", "fused_4_waves", "fused_16_waves", "fused_on_128_entry_segments", "unfused_head",
                              "tile_keys_32_bit", "no_single_launch_sort", "range_sort_all_31_bits", "backward_on_the_slot_grid", "single_launch_sort_up_to_16k",
                              "range_sort_lsd_passes", "range_sort_lsd_all_31_bits", "defaults"])
def test_plan_variants_are_invisible(env, hip_lib_built):
    """The segment plan is an internal choice too: round 1 as the complete walk of the list heads (what the big frames take by default:
    k_render_pass2_grouped<true>, here forced onto the 64-entry plan with heads of 5 and 2 segments), pass 2 over groups of segments
    behind a head, no head on 128-entry segments, the sort's block size and digit width -- the image and the gradients must not
    depend on any of it.  Round 3: the fused one-launch forward against the five-launch form, 16- against 32-bit tile keys, the single-launch
    small sort against the general one (by default up to 4096 pairs; with `LIDARGS_SMALL_SORT_MAX=16384` the third scene's 9000 Gaussians
    go through it too), the range sort on the key span against all 31 bits (the third scene sits closer than 2 m to the sensor in places,
    so that the key span is not the usual 26 bits); the backward blend over
    the slot grid against the work list the combine fills (the default whenever the five-launch forward runs: `five_launch_forward`)."""
    import os, subprocess, sys
    code = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import lidargs_scenes as sc
from util import hip_forward_backward, oracle_forward_backward, parity, GRAD_KEYS_SR
for kind, P, H, W, seed in (("street", 60000, 32, 800, 11), ("shell", 20000, 16, 512, 12), ("near", 9000, 16, 400, 13)):
    if kind == "near":
        scene = sc.make_scene("shell", P, H, seed, random_view=True)
        scene["means3D"] = (scene["means3D"] * np.float32(0.08)).astype(np.float32)      # ranges 0.4 .. 4.8 m
        scene["scales"] = (scene["scales"] * np.float32(0.1)).astype(np.float32)
        grads = sc.upstream_grads(H, W, seed)
        ref = oracle_forward_backward(scene, W, H, grads)
        hip = hip_forward_backward(scene, W, H, grads)
        assert int((hip["radii"] != ref["radii"]).sum()) <= 1
        for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
            parity(kind + "." + k, hip[k], ref[k])
        continue
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    grads = sc.upstream_grads(H, W, seed)
    ref = oracle_forward_backward(scene, W, H, grads)
    hip = hip_forward_backward(scene, W, H, grads)
    for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
        parity(kind + "." + k, hip[k], ref[k])
print("OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (root, os.path.join(root, "lidar-gs_amd"), os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "OK" in r.stdout


@pytest.mark.parametrize("H,W", [(16, 4128), (272, 512)], ids=["258_tile_columns", "272_rows"])
def test_images_too_big_for_compact_span_records(H, W, hip_lib_built):
    """The per-Gaussian span record the tile lists are built from is ONE 32-bit word while the image has at most 256 tile columns
    and 256 rows (csrc/lidargs_common.h span_pack), and the 16-byte form beyond: both forms against the oracle (every other test
    of this file runs the compact one)."""
    from util import hip_forward_backward, oracle_forward_backward, GRAD_KEYS_SR
    scene = sc.make_scene("shell", 15000, H, 77, random_view=True)
    grads = sc.upstream_grads(H, W, 77)
    ref = oracle_forward_backward(scene, W, H, grads)
    hip = hip_forward_backward(scene, W, H, grads)
    assert np.array_equal(hip["radii"], ref["radii"])
    for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
        parity(k, hip[k], ref[k])


def test_device_side_counters_are_counted_inside_the_forward_and_only_on_request(hip_lib_built):
    """lidargs_last_counters (round-5 verdict item 9): V, R_ref, taken instances, touched Gaussians and the backward's entries are counted
    by launches queued at the end of the forward while lidargs_counters_enable(1) is in force, into a page the library owns; the query
    touches none of the caller's memory -- it is made here after every tensor of the frame has been released and the allocator emptied."""
    import gc
    import torch
    from diff_lidargs_rasterization import _C
    scene = sc.make_scene("street", 30000, 64, 29, random_view=True)
    grads = sc.upstream_grads(64, 900, 29)
    hip = hip_forward_backward(scene, 900, 64, grads)
    off = _C.last_counters()
    assert off["P"] == 30000 and off["instances"] > 0 and off["tile_rows"] in (4, 8, 16, 32)
    assert off["V"] == -1 and off["R_ref"] == -1 and off["touched"] == -1 and off["backward_entries"] == -1
    _C.counters_enable(True)
    try:
        hip = hip_forward_backward(scene, 900, 64, grads)
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()       # the frame's buffers are gone
        junk = torch.full((64 << 20,), 255, dtype=torch.uint8, device="cuda"); del junk
        on = _C.last_counters()
    finally:
        _C.counters_enable(False)
    ref = oracle_forward_backward(scene, 900, 64)
    assert on["V"] == int((hip["radii"] > 0).sum())
    assert on["R_ref"] == int(ref["fwd"].num_rendered)                        # the reference's 16x1 instance count
    touched = int((np.abs(hip["dL_dopacity"]).reshape(-1) > 0).sum())
    assert 0 < touched <= on["touched"] <= on["V"]
    assert 0 < on["backward_entries"] <= on["taken_instances"] <= on["instances"] * 8
    assert _C.last_counters() == on                                          # a second query: same numbers, no new work


def test_adaptive_tile_height_is_chosen_and_invisible(hip_lib_built):
    """Tall footprints (scale_modifier 6 on a 64-beam view) make the adaptive choice leave the default 4-row tiles;
    the results must still match the oracle, which knows nothing about tile heights."""
    from diff_lidargs_rasterization import _C
    scene = sc.make_scene("street", 20000, 64, 23, random_view=True)
    grads = sc.upstream_grads(64, 600, 23)
    ref = oracle_forward_backward(scene, 600, 64, grads, scale_modifier=6.0)
    hip = hip_forward_backward(scene, 600, 64, grads, scale_modifier=6.0)
    rows = _C.last_counters()["tile_rows"]
    print("adaptive tile_rows =", rows)
    assert rows in (8, 16, 32), rows
    for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
        parity(k, hip[k], ref[k])


@pytest.mark.parametrize("seed", range(12))
def test_random_small_scenes(seed, hip_lib_built):
    """A sweep of small random scenes (ragged image sizes, random views, both scene kinds, near/far culls) against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    H = int(rng.choice([2, 3, 16, 17, 32, 40, 64]))
    W = int(rng.integers(1, 700))
    P = int(rng.integers(1, 6000))
    kind = "shell" if seed % 2 else "street"
    scene = sc.make_scene(kind, P, H, 50 + seed, random_view=bool(seed % 3))
    grads = sc.upstream_grads(H, W, 50 + seed)
    kw = dict(far=int(rng.choice([80, 30])), near=int(rng.choice([0, 2])), scale_modifier=float(rng.choice([1.0, 0.5, 2.5])))
    ref = oracle_forward_backward(scene, W, H, grads, **kw)
    hip = hip_forward_backward(scene, W, H, grads, **kw)
    assert (hip["radii"] == ref["radii"]).mean() > 0.999
    for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
        parity(k, hip[k], ref[k])


@pytest.mark.parametrize("H,W,P,kind,seed,kw", [(2, 25, 2033, "street", 17, dict(far=30, near=0, scale_modifier=0.5)),
                                                 (40, 6, 4305, "shell", 70, dict(far=30, near=2, scale_modifier=1.0)),
                                                 (17, 22, 3070, "street", 130, dict(far=80, near=2, scale_modifier=1.0)),
                                                 (16, 31, 5000, "shell", 5, dict(far=80, near=0, scale_modifier=1.0))],
                         ids=["2x25", "40x6", "17x22", "16x31"])
def test_narrow_images_where_a_tile_spans_half_the_panorama(H, W, P, kind, seed, kw, hip_lib_built):
    """The reference evaluates a Gaussian at every pixel of the 16-column tiles its rect touches, and a pixel that looks the OTHER way
    (azimuth difference near pi) projects onto the Gaussian's tangent plane at (0, 0): it is blended at full weight there
    (R3/cr/forward.cu:593-606).  With W <= 32 one tile spans that far.  The footprint pruning of the preprocess (a small-angle argument)
    dropped those pixels' entries until tools/parity_sweep.py found it (round 3): 4 of 100 colour entries off by their whole value on
    the first of these scenes."""
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    grads = sc.upstream_grads(H, W, seed)
    ref = oracle_forward_backward(scene, W, H, grads, **kw)
    hip = hip_forward_backward(scene, W, H, grads, **kw)
    assert np.array_equal(hip["radii"], ref["radii"])
    for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
        parity(k, hip[k], ref[k])


def test_rect_upper_tile_edge_takes_two_roundings(hip_lib_built):
    """`getRect_lidar`'s upper column bound is `(int)((p.x + rx + BLOCK_X - 1) / BLOCK_X)` (R3/cr/auxiliary.h:88), evaluated left to
    right in fp32: `+ 16` and `- 1` are two roundings.  For p.x two ulps under column 16 with rx = 1 the first one ties up to 33
    (32.999998 is not a float), so the rect reaches tile 1 and the Gaussian is blended into column 16, the pixel nearest to its centre;
    `+ 15.f` in one step gives 31.999998 and leaves tile 1 out.  The same happens where p.x + rx + 16 crosses 64, 128, ...
    tools/dist_sweep.py found it on one Gaussian of one frame in ~21 k random scenes (round 3, seed 500751); this scene puts a mean
    there by construction."""
    H, W = 16, 31
    scene = sc.make_scene("shell", 4, H, 3, random_view=False)
    scene["means3D"][2, 1] = np.float32(-4.874938011169434)
    grads = sc.upstream_grads(H, W, 3)
    ref = oracle_forward_backward(scene, W, H, grads)
    p_c, p_r = ref["fwd"].array("means2D").reshape(-1, 2)[2]
    assert p_c == np.float32(15.999998092651367) and tuple(ref["fwd"].array("radii_xy").reshape(-1, 2)[2]) == (1, 1)
    row = int(round(float(p_r)))
    assert ref["occ"][0, row, 16] > 0.3, "the Gaussian must reach column 16 on the reference side"
    hip = hip_forward_backward(scene, W, H, grads)
    assert np.array_equal(hip["radii"], ref["radii"])
    assert abs(hip["occ"][0, row, 16] - ref["occ"][0, row, 16]) < 1e-5      # one pixel: inside any outlier budget, so checked by name
    for k in ("color", "depth", "occ") + GRAD_KEYS_SR:
        parity(k, hip[k], ref[k])


def test_non_contiguous_inputs_and_wrong_dtypes(hip_lib_built):
    """The reference binding calls `.contiguous().data<float>()` on every input (R3/rasterize_points.cu:64-90): strided views are
    accepted (same image bit for bit, gradients arrive in the views' own layout), anything but float32 is refused."""
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import make_settings, to_torch
    P, H, W, seed = 6000, 16, 512, 41
    scene = sc.make_scene("street", P, H, seed, random_view=True)
    st = to_torch(scene)
    rast = GaussianRasterizer(make_settings(st, W, H))
    gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed))

    def run(m3, col, op, scl, rot):
        m2 = torch.zeros((P, 4), device="cuda", requires_grad=True)
        color, depth, occ, radii = rast(means3D=m3, means2D=m2, opacities=op, colors_precomp=col, scales=scl, rotations=rot)
        torch.autograd.backward([color, depth, occ], [gc, gd, go])
        return color.detach(), depth.detach(), occ.detach(), radii

    plain = [st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")]
    ref = run(*plain)
    wide = torch.zeros((P, 9), device="cuda"); wide[:, 1:4] = st["means3D"]; wide[:, 5:8] = st["scales"]
    wide.requires_grad_(True)
    colT = st["colors"].t().contiguous().requires_grad_(True)                       # [2, P]: its transpose is a strided [P, 2] view
    rot2 = torch.zeros((2 * P, 4), device="cuda"); rot2[::2] = st["rotations"]; rot2.requires_grad_(True)
    op = st["opacities"].clone().requires_grad_(True)
    got = run(wide[:, 1:4], colT.t(), op, wide[:, 5:8], rot2[::2])
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # float atomics order the sums differently from launch to launch: compare at the parity tolerance
    parity("d means3D (strided view)", wide.grad[:, 1:4].cpu().numpy(), plain[0].grad.cpu().numpy())
    parity("d scales (strided view)", wide.grad[:, 5:8].cpu().numpy(), plain[3].grad.cpu().numpy())
    parity("d colors (transposed)", colT.grad.t().cpu().numpy(), plain[1].grad.cpu().numpy())
    parity("d rotations (every other row)", rot2.grad[::2].cpu().numpy(), plain[4].grad.cpu().numpy())
    assert float(rot2.grad[1::2].abs().max()) == 0.0 and float(wide.grad[:, 0].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="Float"):
        rast(means3D=st["means3D"].double(), means2D=torch.zeros((P, 4), device="cuda"), opacities=st["opacities"], colors_precomp=st["colors"],
             scales=st["scales"], rotations=st["rotations"])


def test_backward_twice_on_one_forward(hip_lib_built):
    """retain_graph: the forward pre-zeroes the per-Gaussian gradient lines for ONE backward; a second backward on the same
    buffers has to start from zero as well (same gradients, not doubled), also when another forward ran in between."""
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import to_torch, make_settings
    H, W = 32, 500
    scene = sc.make_scene("street", 8000, H, 31, random_view=True)
    st = to_torch(scene)
    leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((8000, 4), device="cuda", requires_grad=True)
    rast = GaussianRasterizer(make_settings(st, W, H))
    gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, 31))
    call = lambda: rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                        scales=leaves["scales"], rotations=leaves["rotations"])
    color, depth, occ, _ = call()
    inputs = list(leaves.values()) + [means2D]
    first = torch.autograd.grad([color, depth, occ], inputs, [gc, gd, go], retain_graph=True)
    other = call()                                       # a forward on other buffers in between
    second = torch.autograd.grad([color, depth, occ], inputs, [gc, gd, go], retain_graph=True)
    third = torch.autograd.grad(list(other[:3]), inputs, [gc, gd, go])
    for a, b, c in zip(first, second, third):
        assert torch.isfinite(a).all()
        # float atomics order the sums differently from launch to launch: compare at the parity tolerance
        parity("second backward", b.cpu().numpy(), a.cpu().numpy(), verbose=False)
        parity("other forward's backward", c.cpu().numpy(), a.cpu().numpy(), verbose=False)


def test_intermediate_gradients_are_optional(hip_lib_built):
    """dL_dcov3D is an output only when the covariance was an input: with scales + rotations the module asks the binding not to
    materialise it (want_cov3D_grad=False).  The binding's own surface still returns it by default, it still matches the oracle, and
    leaving it out changes no other gradient."""
    import torch
    from diff_lidargs_rasterization import _C
    from util import to_torch, oracle_forward_backward
    H, W, P = 32, 600, 20000
    scene = sc.make_scene("street", P, H, 33, random_view=True)
    grads = sc.upstream_grads(H, W, 33)
    st = to_torch(scene)
    empty = torch.empty(0, device="cuda")
    fwd = _C.rasterize_gaussians(st["bg"], st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"], 1.0, empty,
                                 st["viewmatrix"], st["viewmatrix"], H, W, st["beams"], empty, 1, empty, False, 80, 0, False)
    R, radii, geom, binning, img = fwd[0], fwd[4], fwd[5], fwd[6], fwd[7]
    gc, gd, go = (torch.from_numpy(g).cuda() for g in grads)
    bw = lambda **kw: _C.rasterize_gaussians_backward(st["bg"], st["means3D"], radii, st["colors"], st["scales"], st["rotations"], 1.0, empty,
                                                      st["viewmatrix"], st["viewmatrix"], st["beams"], 1.0, 1.0, gc, gd, go, empty, 1, empty,
                                                      geom, R, binning, img, False, **kw)
    full, lean = bw(), bw(want_cov3D_grad=False)
    assert full[4] is not None and tuple(full[4].shape) == (P, 6) and lean[4] is None
    ref = oracle_forward_backward(scene, W, H, grads)
    if "dL_dcov3D" in ref:
        parity("dL_dcov3D (scales + rotations path)", full[4].cpu().numpy(), ref["dL_dcov3D"])
    for i, (a, b) in enumerate(zip(full, lean)):
        if i != 4:
            parity(f"output {i} without dL_dcov3D", b.cpu().numpy(), a.cpu().numpy(), verbose=False)   # (float atomics: two launches)


def test_unused_outputs_have_no_gradient(hip_lib_built):
    """A loss that reads only `color`: autograd hands no gradient for depth / occ (none is materialised), which must equal
    passing zeros for them."""
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import to_torch, make_settings
    H, W = 16, 300
    scene = sc.make_scene("shell", 3000, H, 41)
    st = to_torch(scene)
    leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((3000, 4), device="cuda", requires_grad=True)
    rast = GaussianRasterizer(make_settings(st, W, H))
    gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, 41))
    color, depth, occ, _ = rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                                scales=leaves["scales"], rotations=leaves["rotations"])
    inputs = list(leaves.values()) + [means2D]
    only_color = torch.autograd.grad([color], inputs, [gc], retain_graph=True)
    zeros = torch.autograd.grad([color, depth, occ], inputs, [gc, torch.zeros_like(gd), torch.zeros_like(go)])
    for a, b in zip(only_color, zeros):
        parity("color-only backward", a.cpu().numpy(), b.cpu().numpy(), verbose=False)


@pytest.mark.parametrize("how", ["clone", "save_on_cpu"])
def test_backward_on_cloned_buffers(how, hip_lib_built):
    """the backward rebuilds its view from (P, R, W*H) and the buffers' contents alone.  Every saved tensor
    (the three opaque buffers included) is relocated between forward and backward -- cloned to a new address, or offloaded
    to the host and brought back -- and the backward, run twice, must return what an undisturbed run returns.  Run at two
    tile heights: the height the forward chose travels in `num_rendered`, not in a host table."""
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import to_torch, make_settings
    for H, W, P, seed, mod in ((32, 500, 8000, 33, 1.0), (64, 600, 20000, 23, 6.0)):      # 4-row tiles / adaptive 8- or 16-row tiles
        scene = sc.make_scene("street", P, H, seed, random_view=True)
        st = to_torch(scene)
        leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
        means2D = torch.zeros((P, 4), device="cuda", requires_grad=True)
        rast = GaussianRasterizer(make_settings(st, W, H, scale_modifier=mod))
        gc, gd, go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed))
        call = lambda: rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                            scales=leaves["scales"], rotations=leaves["rotations"])
        inputs = list(leaves.values()) + [means2D]
        plain = torch.autograd.grad(list(call()[:3]), inputs, [gc, gd, go])
        moved = []
        if how == "clone":
            def pack(t):
                moved.append(t.data_ptr())
                return t.clone()
            hooks = torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t.clone())      # a fresh address at every unpack too
        else:
            hooks = torch.autograd.graph.save_on_cpu(pin_memory=False)
        with hooks:
            color, depth, occ, _ = call()
        call()                                                # other forwards in between: nothing of theirs may leak in
        first = torch.autograd.grad([color, depth, occ], inputs, [gc, gd, go], retain_graph=True)
        second = torch.autograd.grad([color, depth, occ], inputs, [gc, gd, go])
        for a, b, c in zip(plain, first, second):
            parity("relocated buffers", b.cpu().numpy(), a.cpu().numpy(), verbose=False)
            parity("relocated buffers, 2nd backward", c.cpu().numpy(), a.cpu().numpy(), verbose=False)


def _enqueue_setup(P=20000, H=32, W=600, seed=71, kind="street"):
    import torch
    from diff_lidargs_rasterization import GaussianRasterizer
    from util import to_torch, make_settings
    scene = sc.make_scene(kind, P, H, seed, random_view=True)
    st = to_torch(scene)
    leaves = {k: st[k].clone().requires_grad_(True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros((P, 4), device="cuda", requires_grad=True)
    grads = [torch.from_numpy(g).cuda() for g in sc.upstream_grads(H, W, seed)]
    make = lambda: GaussianRasterizer(make_settings(st, W, H))
    call = lambda rast: rast(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], colors_precomp=leaves["colors"],
                             scales=leaves["scales"], rotations=leaves["rotations"])
    return make, call, list(leaves.values()) + [means2D], grads


def test_enqueue_only_forward_matches_the_ordinary_one(hip_lib_built):
    """lidargs_forward_enqueue (no host wait: capacity from the caller, counts on the device) must produce the ordinary forward's
    image bit for bit -- the forward has no atomics -- and gradients inside the summation-order band; a capacity that is too
    small must be reported (status words, and a RuntimeError from the next call), never silently rendered."""
    import torch
    make, call, inputs, grads = _enqueue_setup()
    exact = make()
    ref = call(exact)
    g_ref = torch.autograd.grad(list(ref[:3]), inputs, grads)
    eq = make()
    eq.enqueue_only = True
    first = call(eq)                                     # learns capacity and tile height (an ordinary frame)
    assert eq._enqueue.cap is not None and eq._enqueue.frames == 0
    second = call(eq)                                    # enqueue-only
    assert eq._enqueue.frames == 1
    g_eq = torch.autograd.grad(list(second[:3]), inputs, grads)
    for a, b, c in zip(ref, first, second):
        assert torch.equal(a, b) and torch.equal(a, c)
    for a, b in zip(g_ref, g_eq):
        parity("enqueue-only backward", b.cpu().numpy(), a.cpu().numpy(), verbose=False)
    st = eq.enqueue_status()
    assert st["needed"] == st["binned"] and not st["overflow"] and st["capacity"] >= st["needed"] > 0, st
    # too small a capacity: flagged on the device, raised by the next forward, and the frame after that is right again
    eq._enqueue.cap = max(64, (st["needed"] // 3) & ~3)
    eq._enqueue.pending = False                           # (the module would otherwise see the last frame's need and grow first)
    bad = call(eq)
    st2 = eq.enqueue_status()
    assert st2["overflow"] and st2["binned"] < st2["needed"], st2
    assert bool(torch.isfinite(bad[0]).all())            # dropped instances, not garbage
    with pytest.raises(RuntimeError, match="binning"):
        call(eq)
    good = call(eq)
    assert not eq.enqueue_status()["overflow"]
    for a, b in zip(ref, good):
        assert torch.equal(a, b)


def test_enqueue_only_capacity_not_a_multiple_of_four(hip_lib_built):
    """The library rounds the caller's capacity up to the multiple of 4 `num_rendered` can carry and uses that ONE number for the
    emit, the sort, the tile ranges and the overflow test: a capacity one to three short of the need, but whose rounding covers it,
    renders the exact image with no overflow; one whose rounding does not is flagged."""
    import torch
    make, call, inputs, grads = _enqueue_setup()
    exact = make()
    ref = call(exact)
    eq = make()
    eq.enqueue_only = True
    call(eq); call(eq)
    need = eq.enqueue_status()["needed"]
    assert need > 64
    for cap, covered in ((need - 1, ((need - 1 + 3) & ~3) >= need), (((need - 4) & ~3) - 1, False), (need + 1, True)):
        eq._enqueue.cap = cap
        eq._enqueue.pending = False
        out = call(eq)
        st = eq.enqueue_status()
        print(f"[enqueue] need {need} capacity {cap} -> library capacity {st['capacity']}, binned {st['binned']}, overflow {st['overflow']}")
        assert st["capacity"] == (cap + 3) & ~3 and st["capacity"] % 4 == 0
        assert bool(st["overflow"]) == (not covered) and st["binned"] == min(need, st["capacity"])
        if covered:
            for a, b in zip(ref, out):
                assert torch.equal(a, b)
        eq._enqueue.pending = False                      # (do not let the module raise for the deliberately short frame)


def test_hip_graph_capture_of_forward_and_backward(hip_lib_built):
    """The enqueue-only frame is capturable: forward + backward recorded once in a HIP graph and replayed; every replay gives the
    eager image bit for bit and the eager gradients within the summation-order band (the backward's float atomics).  Runs in a
    process of its own: a capture is invalidated by unrelated runtime calls made while it records (another test's pinned host
    memory being released, an event query), and a broken capture must fail this test, not take the session down."""
    import os, subprocess, sys
    code = r"""
import sys, gc
sys.path[:0] = [%r, %r, %r]
import numpy as np, torch
import lidargs_scenes as sc
from test_parity_gpu import _enqueue_setup
from util import parity
make, call, inputs, grads = _enqueue_setup(P=30000, H=64, W=800, seed=72)
# (the eager reference is computed AFTER the replays: an autograd graph built on the default stream and still alive would tie
#  the leaves' AccumulateGrad nodes to that stream, which PyTorch warns breaks the capture of a backward -- and it does)
rast = make()
rast.enqueue_only = True
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):                         # warm-up off the default stream, as graph capture asks
    for _ in range(3):
        for t in inputs:
            t.grad = None
        out = call(rast)
        torch.autograd.backward(list(out[:3]), grads)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
assert rast._enqueue.frames >= 2
for t in inputs:
    t.grad = None
gc.collect()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = call(rast)
    torch.autograd.backward(list(out[:3]), grads)
replays = []
for _rep in range(3):
    graph.replay()
    torch.cuda.synchronize()
    replays.append(([o.clone() for o in out], [t.grad.clone() for t in inputs]))
assert not rast.enqueue_status()["overflow"]
del graph
exact = make()
ref = [t.detach() for t in call(exact)]
g_ref = torch.autograd.grad(list(call(exact)[:3]), inputs, grads)
for outs, gs in replays:
    for a, b in zip(ref, outs):
        assert torch.equal(a, b)                      # image planes and radii: bit for bit, every replay
    for a, b in zip(g_ref, gs):
        parity("graph replay backward", b.cpu().numpy(), a.cpu().numpy(), verbose=False)
print("GRAPH-OK")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (root, os.path.join(root, "lidar-gs_amd"), os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "GRAPH-OK" in r.stdout
