"""Diagnostic: per-surfel intermediates of the surfel backward, HIP vs oracle (run on a GPU box)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from util import surfel_scene, surfel_upstream_grads, oracle_surfel_forward_backward, to_torch
from diff_lidargs_surfel_rasterization import _C

kind, P, H, W, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sc = surfel_scene(kind, P, H, seed)
g = surfel_upstream_grads(H, W, seed)
ref = oracle_surfel_forward_backward(sc, W, H, g)
t = to_torch(sc)
e = torch.empty(0, device="cuda")
R, color, others, radii, pixels, gb, bb, ib = _C.rasterize_gaussians(t["bg"], t["means3D"], t["colors"], t["opacities"], t["scales"], t["rotations"], 1.0, e,
    t["viewmatrix"], torch.eye(4).cuda(), t["beams"], H, W, e, 1, torch.zeros(3).cuda(), False, 80, 0, False)
gc, go = (torch.from_numpy(x).cuda() for x in g)
m2d, dcol, dopa, dm3, dtm, dsh, dsc, drot, depth = _C.rasterize_gaussians_backward(t["bg"], t["means3D"], radii, t["colors"], t["scales"], t["rotations"], 1.0, e,
    t["viewmatrix"], torch.eye(4).cuda(), t["beams"], gc, go, e, 1, torch.zeros(3).cuda(), gb, R, bb, ib, False)
hip = dict(dL_dmeans2D=m2d, dL_dcolors=dcol, dL_dopacity=dopa, dL_dmeans3D=dm3, dL_dtransMat=dtm, dL_dscales=dsc, dL_drotations=drot)
rad_h = radii.cpu().numpy()
print("radii mismatches:", np.nonzero(rad_h != ref["radii"])[0][:20], rad_h[rad_h != ref["radii"]][:20], ref["radii"][rad_h != ref["radii"]][:20])
print("color maxdiff", np.abs(color.cpu().numpy() - ref["color"]).max())
pv = sc["means3D"] @ sc["viewmatrix"].reshape(4, 4)[:3, :3] + sc["viewmatrix"].reshape(4, 4)[3, :3]
rng = np.linalg.norm(pv, axis=1)
for k, v in hip.items():
    h = v.cpu().numpy().reshape(P, -1); r = ref[k].reshape(P, -1)
    scale = np.abs(r).max()
    err = np.abs(h - r) / (np.abs(r) + 1e-3 * scale)
    rows = np.argsort(-err.max(axis=1))[:6]
    print(f"== {k}: scale {scale:.3e}, outliers {(err > 1e-4).sum()}, rows with outliers {(err.max(axis=1) > 1e-4).sum()}")
    for i in rows:
        print(f"   surfel {i} err {err[i].max():.2e} radius {rad_h[i]} range {rng[i]:.2f} scales {sc['scales'][i]} opa {sc['opacities'][i]}\n      hip {h[i]}\n      ref {r[i]}")
