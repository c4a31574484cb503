import sys, time, numpy as np, torch
sys.path[:0]=["/root/repo","/root/repo/lidar-gs_amd","/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch, make_settings
from diff_lidargs_rasterization import GaussianRasterizer, _C
import os
cfgs = [sc.BASELINE_CONFIGS["cfg3"]] if os.environ.get("ONLY3") else [("shell",2000,16,512,1), sc.BASELINE_CONFIGS["cfg3"]]
for cfg in cfgs:
    kind,P,H,W,seed = cfg
    scene = sc.make_scene(kind,P,H,seed); st = to_torch(scene)
    rast = GaussianRasterizer(make_settings(st,W,H))
    m2 = torch.zeros(P,4,device="cuda",requires_grad=True)
    leaves = [st[k].clone().requires_grad_(True) for k in ("means3D","colors","opacities","scales","rotations")]
    gc,gd,go = (torch.from_numpy(g).cuda() for g in sc.upstream_grads(H,W,seed))
    tf=tb=0.0; n=30
    for i in range(n+5):
        if i==5: torch.cuda.synchronize(); t0=time.perf_counter(); tf=tb=0
        a=time.perf_counter()
        c,d,o,r = rast(means3D=leaves[0],means2D=m2,opacities=leaves[2],colors_precomp=leaves[1],scales=leaves[3],rotations=leaves[4])
        b=time.perf_counter()
        torch.autograd.backward([c,d,o],[gc,gd,go])
        e=time.perf_counter()
        tf+=b-a; tb+=e-b
    torch.cuda.synchronize(); tot=(time.perf_counter()-t0)/n
    ms=torch.cuda.memory_stats(); print("device allocs", ms["num_device_alloc"], "frees", ms["num_device_free"], "retries", ms["num_alloc_retries"], "reserved MB", ms["reserved_bytes.all.current"]>>20)
    print(P,"frame %.3f ms; host in fwd call %.3f ms (includes the readback wait), host in backward call %.3f ms"%(tot*1e3, tf/n*1e3, tb/n*1e3))
