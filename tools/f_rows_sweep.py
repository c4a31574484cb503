"""Randomized sweep of the widening rows against their numpy oracles: the anchor decode forward + backward
(random N, k in {4, 5, 6, 8, 10}, model flags, visibility), the image loss (random image sizes), the chamfer nearest-neighbour kernel
(random cloud sizes, bit-exact indices and distances) and the densification statistics.

    python tools/f_rows_sweep.py [first_seed] [n] > profiles/rNN_f_rows_sweep.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import util
from util import parity
from oracle import neural_gaussians as ng, lidar_loss as oloss
import test_neural_gaussians_gpu as TNG
import test_lidar_loss_gpu as TL
import test_chamfer as TC
import test_training_statis as TS
from test_neural_gaussians_cpu import PARAM_KEYS

first = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
failed, counts = [], dict(decode=0, loss=0, chamfer=0, stats=0)
mask_flip_cases = []   # decode cases where one opacity within rounding of 0 lands on the other side of the mask: counted, the rest of the case is not comparable
t0 = time.time()


def attempt(kind, desc, fn):
    counts[kind] += 1
    try:
        fn()
    except AssertionError as e:
        failed.append(dict(kind=kind, failed=str(e)[:300], **desc))
    except Exception as e:
        failed.append(dict(kind=kind, failed="EXCEPTION " + repr(e)[:300], **desc))


for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    # ---- f1 decode
    N = int(rng.integers(1, 30000)); k = int(rng.choice([4, 5, 6, 8, 10])); flags = tuple(bool(x) for x in rng.integers(0, 2, 3))

    def decode():
        p, cam, vis, r2 = TNG.random_case(N, k, seed % 10000, flags)
        f = ng.forward(p, cam, vis)
        M = f["xyz"].shape[0]
        ups = [r2.normal(size=s).astype(np.float32) for s in ((M, 3), (M, 2), (M, 1), (M, 3), (M, 4))]
        g = ng.backward(p, f, *ups)
        flips = int((TNG.run_hip(p, cam, vis)["mask"] != f["mask"]).sum())   # forward only: the upstream gradients are sized by the mask
        assert flips <= max(1, int(1e-5 * f["mask"].size)), f"{flips} mask flips"
        if flips:
            mask_flip_cases.append(seed)
            return
        r = TNG.run_hip(p, cam, vis, ups)
        hid = f["_ctx"]["hid"]["opacity"]
        op_scale = float((np.abs(hid) @ np.abs(p["opacity_W2"]).T + np.abs(p["opacity_b2"])).max()) if hid.size else None
        for key in ("xyz", "color", "opacity", "scaling", "rot", "neural_opacity"):
            parity(key, r[key], f[key], scale=(op_scale if "opacity" in key else None), verbose=False)
        for key in ("anchor_feat", "anchor", "offset", "scaling"):
            parity("d" + key, r["g_" + key], g[key], verbose=False)
        for key in PARAM_KEYS:
            parity("d" + key, r["g_" + key], g[key], rtol=5e-4, verbose=False)
    attempt("decode", dict(seed=seed, N=N, k=k, flags=flags), decode)
    # ---- f2 loss
    H = int(rng.choice([1, 2, 3, 16, 17, 64])); W = int(rng.integers(2, 3000))

    def loss():
        image = rng.random((2, H, W), dtype=np.float32); depth = (rng.random((1, H, W), dtype=np.float32) * 70).astype(np.float32)
        gt = np.stack([(rng.random((H, W)) > 0.2).astype(np.float32), rng.random((H, W), dtype=np.float32),
                       np.cumsum(rng.normal(scale=0.004, size=(H, W)), axis=1).astype(np.float32) + 20.0])
        lam = float(rng.choice([0.0, 0.2, 1.0]))
        ref = oloss.forward_backward(image, depth, gt, lam)
        r = TL.run_hip(image, depth, gt, lam)
        for key in TL.TERMS + ("loss",):
            assert abs(r[key] - ref[key]) <= 2e-5 * abs(ref[key]) + 1e-7, (key, r[key], ref[key])
        parity("g_image", r["g_image"], ref["g_image"], verbose=False); parity("g_depth", r["g_depth"], ref["g_depth"], verbose=False)
    attempt("loss", dict(seed=seed, H=H, W=W), loss)
    # ---- f3 chamfer
    B = int(rng.integers(1, 4)); na = int(rng.integers(1, 6000)); nb = int(rng.integers(1, 6000))

    def chamfer():
        import chamfer_3D
        a, b = TC.clouds(B, na, nb, seed % 1000)
        ref = TC.chamfer3d.forward(a, b)
        ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        d1, d2 = torch.zeros(B, na, device="cuda"), torch.zeros(B, nb, device="cuda")
        i1, i2 = torch.zeros(B, na, dtype=torch.int32, device="cuda"), torch.zeros(B, nb, dtype=torch.int32, device="cuda")
        chamfer_3D.forward(ta, tb, d1, d2, i1, i2)
        assert np.array_equal(i1.cpu().numpy(), ref[2]) and np.array_equal(i2.cpu().numpy(), ref[3]), "indices differ"
        assert np.array_equal(d1.cpu().numpy(), ref[0]) and np.array_equal(d2.cpu().numpy(), ref[1]), "distances differ"
    attempt("chamfer", dict(seed=seed, B=B, n=na, m=nb), chamfer)
    # ---- f4 stats
    Ns = int(rng.integers(1, 50000)); ks = int(rng.choice([4, 5, 6, 8, 10])); pvis = float(rng.choice([0.0, 0.3, 0.7, 1.0]))

    def stats():
        vis = rng.random(Ns) < pvis
        nv = int(vis.sum())
        opacity = (rng.random((nv * ks, 1)) * 2 - 1).astype(np.float32)
        sel = (opacity > 0).reshape(-1); M = int(sel.sum())
        c = dict(vis=vis, opacity=opacity, sel=sel, update_filter=rng.random(M) > 0.5, grad=rng.normal(size=(M, 4)).astype(np.float32))
        for fld, shape in zip(TS.FIELDS, ((Ns, 1), (Ns, 1), (Ns * ks, 1), (Ns * ks, 1))):
            c["before_" + fld] = rng.random(shape).astype(np.float32)
        ref = TS.ots.training_statis({fld: c["before_" + fld] for fld in TS.FIELDS}, c["grad"], opacity, c["update_filter"], sel, vis, ks)
        out = TS._run_hip(c, ks)
        for fld in TS.FIELDS:
            np.testing.assert_allclose(out[fld], ref[fld], rtol=2e-6, atol=1e-6, err_msg=f"{fld}")
    attempt("stats", dict(seed=seed, N=Ns, k=ks, pvis=pvis), stats)
log = util.PARITY_LOG
print(json.dumps({"what": "tools/f_rows_sweep.py: decode / loss / chamfer / statistics kernels against their numpy oracles on random cases",
                  "cases": counts, "seconds": round(time.time() - t0, 1), "first_seed": first, "parity_calls": len(log),
                  "entries_compared": int(sum(s["n"] for s in log)), "soft_entries": int(sum(s.get("soft", 0) for s in log)),
                  "flip_entries": int(sum(s.get("flips", 0) for s in log)), "decode_cases_with_a_mask_flip": mask_flip_cases, "failed": failed}, indent=1))
