"""GPU parity of the fused anchor decode: the product function `neural_gaussians.generate_neural_gaussians`
(C ABI lidargs_ng_*) against (i) the golden vectors produced by executing the reference function + torch autograd, and (ii) the
numpy oracle on larger random cases.  Tolerance: 1e-4 relative fp32 (tests/util.py parity metric)."""
import types

import numpy as np
import pytest

import lidargs_scenes as sc

from oracle import neural_gaussians as ng
from test_neural_gaussians_cpu import PARAM_KEYS, load_case
from util import parity

pytestmark = pytest.mark.gpu
# (no outlier-budget override any more: tests/util.py's soft / flip classes apply as everywhere else)


def build_pc(p, device="cuda"):
    return sc.anchor_model_to_torch(p, device)


def run_hip(p, cam, vis, ups=None):
    import torch
    from neural_gaussians import generate_neural_gaussians
    pc = build_pc(p)
    camera = types.SimpleNamespace(camera_center=torch.from_numpy(np.asarray(cam, np.float32)).cuda(), uid=0)
    vmask = None if vis is None else torch.from_numpy(np.asarray(vis)).cuda()
    xyz, color, opacity, scaling, rot, neural_opacity, mask = generate_neural_gaussians(camera, pc, vmask, is_training=True)
    out = dict(xyz=xyz, color=color, opacity=opacity, scaling=scaling, rot=rot, neural_opacity=neural_opacity)
    res = {k: v.detach().cpu().numpy() for k, v in out.items()}
    res["mask"] = mask.cpu().numpy()
    if ups is not None:
        torch.autograd.backward([xyz, color, opacity, scaling, rot], [torch.from_numpy(u).cuda() for u in ups])
        res.update(g_anchor_feat=pc._anchor_feat.grad.cpu().numpy(), g_anchor=pc._anchor.grad.cpu().numpy(),
                   g_offset=pc._offset.grad.cpu().numpy(), g_scaling=pc.get_scaling.grad.cpu().numpy())
        for name in ng.MLPS:
            seq = getattr(pc, "mlp_" + name)
            res[f"g_{name}_W1"], res[f"g_{name}_b1"] = seq[0].weight.grad.cpu().numpy(), seq[0].bias.grad.cpu().numpy()
            res[f"g_{name}_W2"], res[f"g_{name}_b2"] = seq[2].weight.grad.cpu().numpy(), seq[2].bias.grad.cpu().numpy()
    return res


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decode_matches_reference_golden(tag, hip_lib_built):
    p, cam, vis, exp = load_case(tag)
    ups = [exp["up_" + k] for k in ("xyz", "color", "opacity", "scaling", "rot")]
    r = run_hip(p, cam, vis, ups)
    flips = int((r["mask"] != exp["out_mask"]).sum())
    assert flips == 0, f"{flips} opacity-sign flips"
    for k in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity"):
        parity(k, r[k], exp["out_" + k])
    for k in ("anchor_feat", "anchor", "offset", "scaling"):
        parity("d" + k, r["g_" + k], exp["g_" + k])
    for k in PARAM_KEYS:
        parity("d" + k, r["g_" + k], exp["g_" + k], rtol=5e-4)       # fp32 GEMM accumulation order, see the CPU test


def random_case(N, k, seed, flags=(True, True, True)):
    return sc.make_anchor_model(N, k, seed, flags)


@pytest.mark.parametrize("N,k,flags", [(20000, 6, (True, True, True)), (7001, 10, (False, True, True)), (5000, 4, (True, False, False)),
                                       (3000, 5, (False, False, False)), (3000, 8, (True, True, False))])
def test_decode_matches_oracle_random(N, k, flags, hip_lib_built):
    p, cam, vis, rng = random_case(N, k, 100 + k, flags)
    f = ng.forward(p, cam, vis)
    M = f["xyz"].shape[0]
    ups = [rng.normal(size=s).astype(np.float32) for s in ((M, 3), (M, 2), (M, 1), (M, 3), (M, 4))]
    g = ng.backward(p, f, *ups)
    r = run_hip(p, cam, vis, ups)
    # an opacity within rounding of 0 may land on either side of the mask; such pairs are counted, not tolerated in bulk
    flips = int((r["mask"] != f["mask"]).sum())
    assert flips <= max(1, int(1e-5 * f["mask"].size)), flips
    if flips == 0:
        # tanh(y), y = b2 + sum_j W2[o][j] h[j]: a sum of 32 fp32 products that cancels to ~0 for the opacities near the mask's edge, so
        # the rounding error is relative to the magnitude of the TERMS (as for the surfel distortion): scale = max_o (|b2| + sum |W2 h|)
        hid = f["_ctx"]["hid"]["opacity"]
        op_scale = float((np.abs(hid) @ np.abs(p["opacity_W2"]).T + np.abs(p["opacity_b2"])).max())
        for key in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity"):
            parity(key, r[key], f[key], scale=(op_scale if "opacity" in key else None))
        for key in ("anchor_feat", "anchor", "offset", "scaling"):
            parity("d" + key, r["g_" + key], g[key])
        for key in PARAM_KEYS:
            parity("d" + key, r["g_" + key], g[key], rtol=5e-4)


def test_decode_edge_cases(hip_lib_built):
    import torch
    p, cam, vis, _ = random_case(2000, 6, 7)
    r = run_hip(p, cam, np.zeros(2000, bool))                          # nothing visible
    assert r["xyz"].shape == (0, 3) and r["neural_opacity"].shape == (0, 1) and r["mask"].shape == (0,)
    q = dict(p); q["opacity_b2"] = np.full_like(p["opacity_b2"], -50.0)  # everything masked out
    ups = [np.zeros(s, np.float32) for s in ((0, 3), (0, 2), (0, 1), (0, 3), (0, 4))]
    r = run_hip(q, cam, vis, ups)
    assert r["xyz"].shape == (0, 3) and r["mask"].sum() == 0
    assert (r["g_anchor_feat"] == 0).all() and (r["g_cov_W1"] == 0).all()
    r = run_hip(p, cam, None)                                          # visible_mask=None: every anchor
    f = ng.forward(p, cam, None)
    assert np.array_equal(r["mask"], f["mask"])
    parity("xyz", r["xyz"], f["xyz"])
    # unsupported configurations are refused loudly
    from neural_gaussians import generate_neural_gaussians
    pc = build_pc(p); pc.use_feat_bank = True
    with pytest.raises(NotImplementedError):
        generate_neural_gaussians(types.SimpleNamespace(camera_center=torch.zeros(3).cuda(), uid=0), pc)


def test_camera_center_host_copy_follows_the_tensor(hip_lib_built):
    """generate_neural_gaussians keeps the host copy of a camera's centre on the camera object (no device read -- and no drain of the queue --
    per iteration).  It must notice an in-place write and a replaced tensor."""
    import torch
    from neural_gaussians import generate_neural_gaussians
    p, cam, vis = random_case(3000, 4, 11)[:3]
    pc = build_pc(p)
    camera = types.SimpleNamespace(camera_center=torch.from_numpy(np.asarray(cam, np.float32)).cuda(), uid=0)
    def decode(camera_obj):                                               # [M, 5]: positions and the (camera-dependent) colours
        o = generate_neural_gaussians(camera_obj, pc)
        return torch.cat([o[0], o[1]], 1).detach().cpu().numpy()
    fresh = lambda c: decode(types.SimpleNamespace(camera_center=c.clone(), uid=1))
    a0 = decode(camera)
    assert np.array_equal(a0, decode(camera)) and np.array_equal(a0, fresh(camera.camera_center))       # second call: the kept copy
    camera.camera_center.add_(torch.tensor([3.0, -2.0, 0.5], device="cuda"))               # in-place write: the version moves
    b = decode(camera)
    assert np.array_equal(b, fresh(camera.camera_center))
    assert b.shape != a0.shape or not np.array_equal(b, a0)
    camera.camera_center = torch.from_numpy(np.asarray(cam, np.float32)).cuda()           # another tensor
    assert np.array_equal(decode(camera), a0)


def test_decode_gemm_fed_backward_still_matches(hip_lib_built, monkeypatch):
    """The older backward (lidargs_ng_backward: per-anchor rows + two library GEMMs, LIDARGS_NG_ACT_BUFFERS=1) stays in the ABI:
    same gradients as the matrix-pipe one and the golden case."""
    monkeypatch.setenv("LIDARGS_NG_ACT_BUFFERS", "1")
    p, cam, vis, exp = load_case("a")
    ups = [exp["up_" + k] for k in ("xyz", "color", "opacity", "scaling", "rot")]
    r = run_hip(p, cam, vis, ups)
    for k in ("anchor_feat", "anchor", "offset", "scaling"):
        parity("d" + k, r["g_" + k], exp["g_" + k])
    for k in PARAM_KEYS:
        parity("d" + k, r["g_" + k], exp["g_" + k], rtol=5e-4)


@pytest.mark.parametrize("env", [{"LIDARGS_NG_T16_PASSES": "1"}, {"LIDARGS_NG_BACKWARD_T16": "0"}], ids=["t16_one_launch", "mfma_32x32"])
@pytest.mark.parametrize("case", ["a", "b", "random_k4"])
def test_decode_backward_variants_match(env, case, hip_lib_built, monkeypatch):
    """The backward runs as two launches of k_ng_backward_t16 (16x16x4 tiles, k <= 6) by default.  The same kernel as one launch
    (LIDARGS_NG_T16_PASSES=1) and round 4's k_ng_backward_mfma (32x32x2 tiles, LIDARGS_NG_BACKWARD_T16=0: the path k = 8, 10 take) stay
    selectable: same gradients against the reference's golden cases and the oracle, same partial-row layout for the reduction."""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    if case in ("a", "b"):
        p, cam, vis, exp = load_case(case)
        ups = [exp["up_" + k] for k in ("xyz", "color", "opacity", "scaling", "rot")]
        ref = {k: exp["g_" + k] for k in ("anchor_feat", "anchor", "offset", "scaling") + tuple(PARAM_KEYS)}
    else:
        p, cam, vis, rng = random_case(5000, 4, 13, (True, False, False))
        f = ng.forward(p, cam, vis)
        M = f["xyz"].shape[0]
        ups = [rng.normal(size=s).astype(np.float32) for s in ((M, 3), (M, 2), (M, 1), (M, 3), (M, 4))]
        g = ng.backward(p, f, *ups)
        ref = {k: g[k] for k in ("anchor_feat", "anchor", "offset", "scaling") + tuple(PARAM_KEYS)}
    r = run_hip(p, cam, vis, ups)
    if case == "random_k4" and int((r["mask"] != f["mask"]).sum()):
        pytest.skip("an opacity within rounding of 0 landed on the other side of the mask: the upstream rows do not line up")
    for k in ("anchor_feat", "anchor", "offset", "scaling"):
        parity("d" + k, r["g_" + k], ref[k])
    for k in PARAM_KEYS:
        parity("d" + k, r["g_" + k], ref[k], rtol=5e-4)


def test_forward_tile_forms_agree(hip_lib_built, monkeypatch):
    """The forward's MLPs run on 16x16x4 tiles (k_ng_opacity_t16 / k_ng_decode_t16, k <= 6) by default and on 32x32x2 tiles with
    LIDARGS_NG_FORWARD_T16=0 (the path k = 8, 10 take).  Both accumulate bias first, inputs in ascending order; the matrix pipe adds the
    four products of a 16x16x4 step in its own order, so outputs may differ in the last bit (measured: 1 ulp on a few colours) -- the mask
    (opacity > 0) agrees on these cases and every output agrees to 2e-6; k = 6 with a visibility mask, k = 5 (an odd offset count: the
    halves of an anchor's lane pair own 3 + 2 offsets) and k = 4."""
    for N, k, seed, flags in ((9000, 6, 31, (True, True, True)), (4000, 5, 32, (False, True, False)), (3000, 4, 33, (True, False, True))):
        p, cam, vis, _rng = random_case(N, k, seed, flags)
        monkeypatch.setenv("LIDARGS_NG_FORWARD_T16", "1")
        t16 = run_hip(p, cam, vis, None)
        monkeypatch.setenv("LIDARGS_NG_FORWARD_T16", "0")
        wide = run_hip(p, cam, vis, None)
        assert np.array_equal(t16["mask"], wide["mask"])
        for key in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity"):
            d = float(np.abs(t16[key].astype(np.float64) - wide[key]).max())
            assert d <= 2e-6 * max(1.0, float(np.abs(wide[key]).max())), (k, key, d)

@pytest.mark.parametrize("env", [{}, {"LIDARGS_NG_T16_PASSES": "1"}, {"LIDARGS_NG_BACKWARD_T16": "0"}], ids=["t16_two_launches", "t16_one_launch", "mfma_32x32"])
def test_decode_backward_with_a_gradient_on_neural_opacity(env, hip_lib_built, monkeypatch):
    """A loss on `neural_opacity` itself (the sixth output of the training path: every VISIBLE anchor has work then, selected offsets or
    not) next to the five usual upstream gradients: HIP against float64 autograd of the reference's chain of framework ops
    (oracle/neural_gaussians_torch.py) on the same inputs, for the three forms of the backward; k = 6 with a visibility mask and k = 5."""
    import torch
    from neural_gaussians import generate_neural_gaussians
    from oracle import neural_gaussians_torch as ngt
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    for N, k, seed, flags in ((3000, 6, 41, (True, True, True)), (2000, 5, 42, (True, False, True))):
        p, cam, vis, rng = random_case(N, k, seed, flags)
        pc = build_pc(p)
        camera = types.SimpleNamespace(camera_center=torch.from_numpy(np.asarray(cam, np.float32)).cuda(), uid=0)
        vmask = torch.from_numpy(np.asarray(vis)).cuda()
        outs = generate_neural_gaussians(camera, pc, vmask, is_training=True)
        ws = [torch.from_numpy(rng.normal(size=tuple(o.shape)).astype(np.float32)).cuda() for o in outs[:6]]
        sum((o * w).sum() for o, w in zip(outs[:6], ws)).backward()
        got = {"anchor_feat": pc._anchor_feat.grad, "anchor": pc._anchor.grad, "offset": pc._offset.grad, "scaling": pc.get_scaling.grad}
        for name in ng.MLPS:
            seq = getattr(pc, "mlp_" + name)
            got[f"{name}_W1"], got[f"{name}_b1"], got[f"{name}_W2"], got[f"{name}_b2"] = seq[0].weight.grad, seq[0].bias.grad, seq[2].weight.grad, seq[2].bias.grad
        # the reference's graph in float64 on the CPU
        leaf = lambda a: torch.from_numpy(np.asarray(a, np.float64)).requires_grad_(True)
        L = {n: leaf(p[n]) for n in ("anchor_feat", "anchor", "offset", "scaling")}
        P = {m: tuple(leaf(p[f"{m}_{q}"]) for q in ("W1", "b1", "W2", "b2")) for m in ng.MLPS}
        ref = ngt.generate(L["anchor_feat"], L["anchor"], L["offset"], L["scaling"], P, torch.from_numpy(np.asarray(cam, np.float64)),
                           torch.from_numpy(np.asarray(vis)), flags)
        if not np.array_equal(ref[6].numpy(), outs[6].cpu().numpy()):
            pytest.skip("an opacity within rounding of 0 landed on the other side of the mask: the rows do not line up")
        sum((o * w.double().cpu()).sum() for o, w in zip(ref[:6], ws)).backward()
        for n in ("anchor_feat", "anchor", "offset", "scaling"):
            parity("d" + n, got[n].cpu().numpy(), L[n].grad.numpy())
        for m in ng.MLPS:
            for q, t in zip(("W1", "b1", "W2", "b2"), P[m]):
                parity(f"d{m}_{q}", got[f"{m}_{q}"].cpu().numpy(), t.grad.numpy(), rtol=5e-4)


def test_decode_without_transposed_weights_uses_the_per_lane_kernel(hip_lib_built, monkeypatch):
    """A caller of the C ABI that passes no W2T gets the one-anchor-per-lane decode (k_ng_decode): same outputs as the golden case
    (forward only: the backward needs W2T)."""
    import neural_gaussians as prod
    real = prod._model_struct
    monkeypatch.setattr(prod, "_model_struct", lambda k, flags, params, w2t=None: real(k, flags, params, None))
    p, cam, vis, exp = load_case("a")
    r = run_hip(p, cam, vis, None)
    assert int((r["mask"] != exp["out_mask"]).sum()) == 0
    for k in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity"):
        parity(k, r[k], exp["out_" + k])


def test_decode_queued_behind_the_selection_equals_the_waiting_form(hip_lib_built, monkeypatch):
    """By default the decode is queued right behind the selection with output arrays of N k rows (the upper bound of M) and the host
    waits for the two counts while it runs (`lidargs_ng_forward_select_enqueue`); `LIDARGS_NG_CAPACITY_BYTES=0` (or an N k x 52-byte
    block above the limit) waits first and allocates M rows.  Same kernels on the same inputs: every output, the mask and every
    gradient bit for bit, and the outputs are contiguous [M, c] tensors either way."""
    import neural_gaussians as mod
    p, cam, vis, rng = random_case(9000, 6, 21, (True, True, True))
    f = ng.forward(p, cam, vis)
    M = f["xyz"].shape[0]
    ups = [rng.normal(size=s).astype(np.float32) for s in ((M, 3), (M, 2), (M, 1), (M, 3), (M, 4))]
    monkeypatch.setattr(mod, "_CAPACITY_BYTES", 1 << 30)             # whatever LIDARGS_NG_CAPACITY_BYTES says in this environment
    queued = run_hip(p, cam, vis, ups)
    monkeypatch.setattr(mod, "_CAPACITY_BYTES", 0)
    waited = run_hip(p, cam, vis, ups)
    assert queued["xyz"].shape == (M, 3) and queued["rot"].shape == (M, 4)
    for key in queued:
        assert np.array_equal(queued[key], waited[key]), key
