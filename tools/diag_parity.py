"""Diagnostic: per-tensor parity of every backward intermediate (calls the C ABI directly)."""
import sys, ctypes as C, numpy as np, torch
sys.path[:0]=["/root/repo","/root/repo/lidar-gs_amd","/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch, oracle_forward_backward, parity
from diff_lidargs_rasterization import _C
lib=_C._lib
kind,P,H,W,seed = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv)>5 else ("shell",10000,16,512,1)
scene = sc.make_scene(kind,P,H,seed, random_view=("rv" in sys.argv))
grads = sc.upstream_grads(H,W,seed)
ref = oracle_forward_backward(scene,W,H,grads)
st = to_torch(scene)
n, color, depth, occ, radii, gb, bb, ib = _C.rasterize_gaussians(st["bg"], st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"], 1.0, torch.Tensor([]), st["viewmatrix"], torch.eye(4).cuda(), H, W, st["beams"], torch.Tensor([]), 1, torch.zeros(3).cuda(), False, 80, 0, False)
z=lambda *s: torch.zeros(*s,device="cuda")
g=dict(dL_dmeans2D=z(P,4),dL_dconic=z(P,4),dL_dopacity=z(P,1),dL_dcolors=z(P,2),dL_ddepths=z(P,1),dL_dmeans3D=z(P,3),dL_dsphere=z(P,3),dL_dbasis_u1=z(P,3),dL_dbasis_u2=z(P,3),dL_dcov3D=z(P,6),dL_dscales=z(P,3),dL_drotations=z(P,4))
gc,gd,go=(torch.from_numpy(x).cuda() for x in grads)
p=lambda t: C.c_void_p(t.data_ptr())
rc=lib.lidargs_backward(C.c_int(P),C.c_int(1),C.c_int(0),C.c_int(n),p(st["bg"]),C.c_int(W),C.c_int(H),p(st["means3D"]),None,p(st["colors"]),p(st["scales"]),C.c_float(1.0),p(st["rotations"]),None,p(st["viewmatrix"]),None,None,p(st["beams"]),C.c_float(1),C.c_float(1),p(radii),p(gb),p(bb),p(ib),p(gc),p(gd),p(go),
  p(g["dL_dmeans2D"]),p(g["dL_dconic"]),p(g["dL_dopacity"]),p(g["dL_dcolors"]),p(g["dL_ddepths"]),p(g["dL_dmeans3D"]),p(g["dL_dsphere"]),p(g["dL_dbasis_u1"]),p(g["dL_dbasis_u2"]),p(g["dL_dcov3D"]),None,p(g["dL_dscales"]),p(g["dL_drotations"]),C.c_int(0),None)
torch.cuda.synchronize()
assert rc==0, _C._err()
print("radii mismatch", int((radii.cpu().numpy()!=ref["radii"]).sum()))
for k,a,b in (("color",color,ref["color"]),("depth",depth,ref["depth"]),("occ",occ,ref["occ"])):
    try: parity(k,a.cpu().numpy(),b)
    except AssertionError as e: print("   FAIL",str(e)[:120])
for k in g:
    try: parity(k,g[k].cpu().numpy(),ref[k])
    except AssertionError as e: print("   FAIL",str(e)[:120])
# worst offenders of dL_dmeans3D
a=g["dL_dmeans3D"].cpu().numpy(); b=ref["dL_dmeans3D"]
err=np.abs(a-b)/(np.abs(b)+1e-3*np.abs(b).max()); idx=np.argsort(err.max(1))[-5:]
for i in idx:
    print(i, "hip",a[i],"ref",b[i],"err",err[i], "radii",ref["radii"][i], "dist", np.linalg.norm(scene["means3D"][i]), "scales", scene["scales"][i])
    for k in ("dL_dmeans2D","dL_dconic","dL_dbasis_u1","dL_dbasis_u2","dL_dsphere","dL_ddepths","dL_dcov3D"):
        print("    ",k, g[k][i].cpu().numpy(), ref[k][i])
