"""Diagnostic: per-tile list length vs consumed length (reads the opaque image buffer by its known layout)."""
import sys, numpy as np, torch
sys.path[:0]=["/root/repo","/root/repo/lidar-gs_amd","/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch
from diff_lidargs_rasterization import _C
kind,P,H,W,seed = sc.BASELINE_CONFIGS[sys.argv[1] if len(sys.argv)>1 else "cfg3"]
scene = sc.make_scene(kind,P,H,seed); st = to_torch(scene)
n, color, depth, occ, radii, gb, bb, ib = _C.rasterize_gaussians(st["bg"], st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"], 1.0, torch.Tensor([]), st["viewmatrix"], torch.eye(4).cuda(), H, W, st["beams"], torch.Tensor([]), 1, torch.zeros(3).cuda(), False, 80, 0, False)
cnt=_C.last_counters(); TH=cnt["tile_rows"]; tiles=cnt["tiles"]; N=H*W
al=lambda x:(x+127)//128*128
o_nc=al(4*N); o_tp=al(o_nc+4*N); o_rg=al(o_tp+4*N)
buf=ib.cpu().numpy()
ncontrib=buf[o_nc:o_nc+4*N].view(np.uint32).reshape(H,W)
finalT=buf[0:4*N].view(np.float32).reshape(H,W)
ranges=buf[o_rg:o_rg+8*tiles].view(np.uint32).reshape(tiles,2)
L=(ranges[:,1]-ranges[:,0]).astype(np.int64)
tx=(W+15)//16
# per-tile consumed = max n_contrib over the tile's pixels if all pixels done early, else full list
cons=np.zeros(tiles,np.int64); unsat=np.zeros(tiles,bool)
for t in range(tiles):
    ty,txx=divmod(t,tx)
    blk=ncontrib[ty*TH:(ty+1)*TH, txx*16:(txx+1)*16]; T=finalT[ty*TH:(ty+1)*TH, txx*16:(txx+1)*16]
    cons[t]=blk.max(); 
print(cnt)
print("list len: mean %.0f median %.0f p90 %.0f p99 %.0f max %d sum %d"%(L.mean(),np.median(L),np.quantile(L,.9),np.quantile(L,.99),L.max(),L.sum()))
print("last contributor (bwd walk): mean %.0f median %.0f p90 %.0f p99 %.0f max %d sum %d"%(cons.mean(),np.median(cons),np.quantile(cons,.9),np.quantile(cons,.99),cons.max(),cons.sum()))
print("n_contrib per pixel mean %.1f max %d ; finalT<2e-4 frac %.3f ; occ mean %.3f"%(ncontrib.mean(), ncontrib.max(), (finalT<2e-4).mean(), 1-finalT.mean()))
rows=L.reshape(-1,tx)
print("per tile-row list len mean:", rows.mean(1).astype(int))
print("per tile-row last-contrib mean:", cons.reshape(-1,tx).mean(1).astype(int))
