"""Row f4, second half: anchor growing (GaussianModel.anchor_growing, /root/reference/scene/gaussian_model.py:677-775).

CPU: the numpy oracle against golden vectors from EXECUTING the reference method (tests/golden/make_anchor_growing_golden.py) -- bit for
bit, every parameter tensor of the grown model.  GPU: the native call through the drop-in function against the same golden vectors
(IEEE quotient, what the CPU-executed reference computes), against the oracle at 1.2 M anchors x 6 offsets in both quotient conventions,
and against the reference's op sequence run by torch ON THE DEVICE (the default convention = what the reference computes where it runs).
Integer work: every comparison is exact (anchor sets, row order, features)."""
import os
import types

import numpy as np
import pytest

from oracle import anchor_growing as oag

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "anchor_growing_golden.npz")
PARAMS = ("anchor", "offset", "anchor_feat", "opacity", "scaling", "rotation")
STATS = ("anchor_demon", "opacity_accum")


def load(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}


def initial_state(c):
    N = int(c["N"])
    return dict(anchor=c["in_anchor"], offset=c["in_offset"], scaling=c["in_scaling"], anchor_feat=c["in_anchor_feat"],
                rotation=np.tile(np.array([[1, 0, 0, 0]], np.float32), (N, 1)), opacity=np.full((N, 1), 0.25, np.float32),
                anchor_demon=np.arange(N, dtype=np.float32).reshape(N, 1), opacity_accum=np.arange(N, dtype=np.float32).reshape(N, 1) * 0.5)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_oracle_matches_reference_execution(tag):
    c = load(tag)
    rands = [c["rand%d" % i] for i in range(int(c["n_rand"]))]
    assert len(rands) == 3                                  # the random numbers are drawn at every level, skipped or not (:687 sits above :691-693)
    s, counts = oag.anchor_growing(initial_state(c), c["grads"], float(c["threshold"]), c["offset_mask"], rands, float(c["voxel_size"]))
    for n in PARAMS + STATS:
        assert s[n].shape == c["out_" + n].shape, (n, s[n].shape, c["out_" + n].shape)
        assert np.array_equal(s[n], c["out_" + n]), n
    if tag == "c":
        assert counts[1] == (-1, -1, -1) and counts[2] == (-1, -1, -1)      # level 0 grew nothing -> the finer levels never run
    else:
        assert all(u > 0 for (_, _, u) in counts)
        assert any(v > u for (_, v, u) in counts)            # some candidate voxels already held an anchor
        assert any(cc > v for (cc, v, _) in counts)          # some voxels received several candidates (the feature maximum)


def test_the_two_quotient_conventions_are_distinguishable():
    """x / s (torch CPU) and x * (1 / s) (torch device kernels) put some offsets of case `a` into different voxels: the fixture pins the former,
    the GPU suite checks the latter against torch on the device."""
    c = load("a")
    rands = [c["rand%d" % i] for i in range(3)]
    s, _ = oag.anchor_growing(initial_state(c), c["grads"], float(c["threshold"]), c["offset_mask"], rands, float(c["voxel_size"]), exact_division=False)
    assert s["anchor"].shape != c["out_anchor"].shape or not np.array_equal(s["anchor"], c["out_anchor"])


# ------------------------------------------------------------------------------------------------ GPU
class _Model:
    """The attributes the drop-in function reads and replaces, with a plain concatenating `cat_tensors_to_optimizer` (the optimizer surgery
    is the model's own code and stays the reference's: out of scope)."""

    def __init__(self, c, dev="cuda"):
        import torch
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        s = initial_state(c)
        self.n_offsets, self.voxel_size = int(c["k"]), float(c["voxel_size"])
        self.update_depth, self.update_init_factor, self.update_hierachy_factor = 3, 16, 4
        for n in PARAMS:
            setattr(self, "_" + n, t(s[n]))
        self.anchor_demon, self.opacity_accum = t(s["anchor_demon"]), t(s["opacity_accum"])

    get_anchor = property(lambda self: self._anchor)

    @property
    def get_scaling(self):
        import torch
        return 1.0 * torch.exp(self._scaling)

    def cat_tensors_to_optimizer(self, d):
        import torch
        return {n: torch.cat((getattr(self, "_" + n), d[n]), dim=0) for n in PARAMS}

    def prune_anchor(self, mask):
        """What the reference's prune_anchor / _prune_anchor_optimizer leave in the parameters (:625-675), optimizer state aside: the kept rows,
        and -- a quirk of :644-648 -- columns 3.. of the (log-space) scaling clamped at 0.05."""
        keep = ~mask
        for n in PARAMS:
            setattr(self, "_" + n, getattr(self, "_" + n)[keep])
        tail = self._scaling[:, 3:]
        tail[tail > 0.05] = 0.05


def _fixed_rands(monkeypatch, rands):
    import torch
    it = iter(rands)
    monkeypatch.setattr(torch, "rand_like", lambda t, *a, **k: torch.from_numpy(next(it)).to(t.device))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_hip_matches_reference_golden(tag, hip_lib_built, monkeypatch):
    import torch
    import anchor_growing as ag
    c = load(tag)
    m = _Model(c)
    _fixed_rands(monkeypatch, [c["rand%d" % i] for i in range(3)])
    ag.anchor_growing(m, torch.from_numpy(c["grads"]).cuda(), float(c["threshold"]), torch.from_numpy(c["offset_mask"]).cuda(), flags=ag.EXACT_DIVISION)
    for n in PARAMS:
        got = getattr(m, "_" + n).cpu().numpy()
        assert got.shape == c["out_" + n].shape, (n, got.shape, c["out_" + n].shape)
        if n in ("scaling", "opacity"):      # rows of the new anchors: log(cur_size), log(0.9 / 0.1) by the DEVICE's logf (constant fills by framework ops, not the native call's output)
            np.testing.assert_allclose(got, c["out_" + n], rtol=1e-6)
        else:
            assert np.array_equal(got, c["out_" + n]), n
    for n in STATS:
        assert np.array_equal(getattr(m, n).cpu().numpy(), c["out_" + n]), n


@pytest.mark.gpu
def test_adjust_anchor_matches_reference_golden(hip_lib_built, monkeypatch):
    """GaussianModel.adjust_anchor as a whole (:776-830) -- gradient norms from the accumulators (0 / 0 -> NaN -> 0), the growing, the
    statistics' reset and padding, the prune masks, the pruned rows -- against the executed reference method (case `adj` of the fixture)."""
    import torch
    import anchor_growing as ag
    c = load("adj")
    c["N"], c["k"] = int(c["N"]), int(c["k"])
    m = _Model(c)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for n in ("offset_gradient_accum", "offset_denom", "anchor_demon", "opacity_accum"):
        setattr(m, n, t(c["in_" + n]))
    _fixed_rands(monkeypatch, [c["rand%d" % i] for i in range(3)])
    ag.adjust_anchor(m, check_interval=100, success_threshold=0.1, grad_threshold=float(c["threshold"]), min_opacity=0.005, flags=ag.EXACT_DIVISION)
    for n in PARAMS:
        got = getattr(m, "_" + n).cpu().numpy()
        assert got.shape == c["out_" + n].shape, (n, got.shape, c["out_" + n].shape)
        if n in ("scaling", "opacity"):
            np.testing.assert_allclose(got, c["out_" + n], rtol=1e-6)
        else:
            assert np.array_equal(got, c["out_" + n]), n
    for n in ("offset_gradient_accum", "offset_denom", "anchor_demon", "opacity_accum", "max_radii2D"):
        got = getattr(m, n).cpu().numpy()
        assert got.shape == c["out_" + n].shape and np.array_equal(got, c["out_" + n]), n


def _big_case(N, k, seed, voxel=0.01):
    import lidargs_scenes as sc
    return sc.anchor_scene(N, k, seed, voxel)


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [0, 1])
def test_hip_matches_oracle_at_training_size(exact, hip_lib_built):
    """1.2 M anchors x 6 offsets (the size the round-5 brief names), the three levels' parameters, each level on the same anchors."""
    import torch
    import anchor_growing as ag
    c = _big_case(1_200_000, 6, 21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    act = np.exp(c["scaling"]).astype(np.float32)
    rng = np.random.default_rng(5)
    for level in range(3):
        thr, rthr, size = oag.level_parameters(level, 0.0005, c["voxel"])
        rand = rng.random(c["N"] * c["k"]).astype(np.float32)
        cand = oag.candidate_mask(c["grads"], c["offset_mask"], rand, thr, rthr, c["N"] * c["k"])
        ra, rf, n_c, n_v = oag.grow_level(c["anchor"], c["offset"], act, c["feat"], cand, size, exact_division=bool(exact))
        ga, gf, counts = ag.grow_level(t(c["anchor"]), t(c["offset"]), t(act), t(c["feat"]), t(c["grads"]), t(c["offset_mask"]), t(rand), thr, rthr, size,
                                       flags=ag.EXACT_DIVISION if exact else 0)
        assert counts == (n_c, n_v, ra.shape[0]), (level, counts, n_c, n_v, ra.shape)
        assert n_c > (100_000, 10_000, 100)[level] and n_v >= ra.shape[0] > 0 and (level == 2 or n_v > ra.shape[0])
        assert np.array_equal(ga.cpu().numpy(), ra), level
        assert np.array_equal(gf.cpu().numpy(), rf), level


@pytest.mark.gpu
def test_default_convention_is_what_torch_computes_on_the_device(hip_lib_built):
    """The reference's op sequence run by torch on this GPU (oracle/anchor_growing_torch.py) against the native call with default flags:
    same anchors, same order, same features -- including anchors grown by an earlier level (N > N0) and a voxel size that is no power of two."""
    import torch
    import anchor_growing as ag
    from oracle import anchor_growing_torch as agt
    c = _big_case(60_000, 6, 22, voxel=0.013)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    anchor, offset, scaling, feat = t(c["anchor"]), t(c["offset"]), torch.exp(t(c["scaling"])), t(c["feat"])
    grads, om = t(c["grads"]), t(c["offset_mask"])
    n0 = c["N"]
    g = torch.Generator(device="cuda").manual_seed(3)
    differs_from_exact = False
    for level in range(3):
        thr, rthr, size = oag.level_parameters(level, 0.0005, c["voxel"])
        rand = torch.rand(n0 * c["k"], device="cuda", generator=g)
        ra, rf, rc = agt.grow_level(anchor, offset, scaling, feat, grads, om, rand, thr, rthr, size, c["k"])
        ga, gf, gc = ag.grow_level(anchor, offset, scaling, feat, grads, om, rand, thr, rthr, size, n_initial=n0)
        assert gc == rc and rc[2] > 0, (level, gc, rc)
        assert torch.equal(ga, ra) and torch.equal(gf, rf), level
        ea, _, _ = ag.grow_level(anchor, offset, scaling, feat, grads, om, rand, thr, rthr, size, n_initial=n0, flags=ag.EXACT_DIVISION)
        differs_from_exact |= ea.shape != ga.shape or not torch.equal(ea, ga)
        U = ga.shape[0]                                            # grow, as the method does (:734-769)
        anchor = torch.cat([anchor, ga]); feat = torch.cat([feat, gf])
        offset = torch.cat([offset, torch.zeros(U, c["k"], 3, device="cuda")])
        scaling = torch.cat([scaling, torch.full((U, 6), size, device="cuda")])
    assert differs_from_exact        # the case can tell the conventions apart (otherwise the test above would prove nothing about the default)


@pytest.mark.gpu
def test_degenerate_inputs(hip_lib_built):
    import torch
    import anchor_growing as ag
    c = _big_case(5000, 4, 23)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    act = torch.exp(t(c["scaling"]))
    args = (t(c["anchor"]), t(c["offset"]), act, t(c["feat"]))
    # no candidate at all
    a, f, counts = ag.grow_level(*args, t(c["grads"]), t(c["offset_mask"]), None, 1e9, 0.5, 0.16)
    assert a.shape == (0, 3) and f.shape == (0, 32) and counts == (0, 0, 0)
    # every candidate falls into a voxel that holds an anchor: zero offsets, every offset a candidate
    z = torch.zeros_like(args[1])
    a, f, counts = ag.grow_level(args[0], z, act, args[3], torch.ones(5000 * 4, device="cuda"), torch.ones(5000 * 4, dtype=torch.bool, device="cuda"), None, 0.5, 0.5, 0.01)
    assert a.shape == (0, 3) and counts[0] == 20000 and counts[2] == 0
    # one huge voxel: everything lands in voxel (0, 0, 0), which holds anchors -> nothing; after moving the anchors away -> exactly one, features = column maxima
    far = args[0] + 1e4
    sel = torch.ones(5000 * 4, dtype=torch.bool, device="cuda")
    off = (args[1] * 0 + 1.0)
    a, f, counts = ag.grow_level(far, off - 1e4 / act[:, None, :3], act, args[3], torch.ones(5000 * 4, device="cuda"), sel, None, 0.5, 0.5, 4096.0)
    assert counts == (20000, 1, 1) and torch.equal(a, torch.zeros(1, 3, device="cuda"))
    assert torch.equal(f[0], args[3].max(0).values)
    # inputs at addresses that are not 16-byte aligned (views into larger tensors) take the kernel's scalar loads: same anchors
    g0, m0 = t(c["grads"]), t(c["offset_mask"])
    rnd = torch.rand(5000 * 4, device="cuda")
    ref_a, ref_f, ref_c = ag.grow_level(*args, g0, m0, rnd, 0.0005, 0.5, 0.16)
    shift = lambda x: torch.cat([x[:1], x])[1:]
    a, f, counts = ag.grow_level(*args, shift(g0), shift(m0), shift(rnd), 0.0005, 0.5, 0.16)
    assert shift(g0).data_ptr() % 16 != 0 and counts == ref_c and ref_c[2] > 0 and torch.equal(a, ref_a) and torch.equal(f, ref_f)
    # CPU tensors are refused (no CPU path)
    with pytest.raises(RuntimeError):
        ag.grow_level(args[0].cpu(), args[1].cpu(), act.cpu(), args[3].cpu(), t(c["grads"]).cpu(), t(c["offset_mask"]).cpu(), None, 0.1, 0.5, 0.16)
    # key width: voxel coordinates spanning more than 63 bits in total are refused, not mis-hashed
    wide = args[0].clone()
    wide[0] = torch.tensor([3e9, -3e9, 3e9]); wide[1] = torch.tensor([-3e9, 3e9, -3e9])
    with pytest.raises(RuntimeError, match="63 key bits"):
        ag.grow_level(wide, z, act, args[3], torch.ones(5000 * 4, device="cuda"), sel, None, 0.5, 0.5, 1.0)
