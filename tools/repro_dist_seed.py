import os, sys
ROOT='/root/repo'
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import lidargs_scenes as sc
from util import hip_forward_backward, oracle_forward_backward
seed=int(sys.argv[1])
rng = np.random.default_rng(seed)
H = int(rng.choice([2, 3, 16, 17, 32, 40, 64])); W = int(rng.choice([int(rng.integers(1, 40)), int(rng.integers(40, 900))]))
P = int(rng.integers(50, 20000))
world = int(rng.choice([2, 3, 4, 5, 8])); wedges = bool(rng.integers(0, 2)) and (W + 15) // 16 >= world
kind = "shell" if rng.random() < 0.5 else "street"
beams = str(rng.choice(["uniform", "waymo", "neartie"])) if H >= 4 else "uniform"
sync = str(rng.choice(["all_reduce", "reduce_scatter"]))
scene = sc.make_scene(kind, P, H, seed % 1000, random_view=True, beams=beams)
scene["bg"] = np.array([rng.random() * 0.5, rng.random() * 0.5], np.float32) if rng.random() < 0.5 else scene["bg"]
print(dict(H=H,W=W,P=P,world=world,wedges=wedges,kind=kind,beams=beams,bg=scene["bg"]))
hip = hip_forward_backward(scene, W, H, None); ref = oracle_forward_backward(scene, W, H, None)
for k in ("color","depth","occ"):
    d=np.abs(hip[k]-ref[k]); bad=np.argwhere(d>1e-3*(np.abs(ref[k])+1e-3*np.abs(ref[k]).max()))
    print(k,"bad",len(bad),"max",d.max(),[tuple(int(x) for x in b) for b in bad[:8]], [float(hip[k][tuple(b)]) for b in bad[:4]], [float(ref[k][tuple(b)]) for b in bad[:4]])
print("radii mism", int((hip["radii"]!=ref["radii"]).sum()))
