"""Diagnostic: how many list entries pass 1 visits vs how many it flags (some pixel of the patch takes them)."""
import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/lidar-gs_amd", "/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch
from diff_lidargs_rasterization import _C
_C.counters_enable(True)      # diagnostics tool: every forward ends with the counting launches of last_counters()
kind, P, H, W, seed = sc.BASELINE_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
scene = sc.make_scene(kind, P, H, seed); st = to_torch(scene)
out = _C.rasterize_gaussians(st["bg"], st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"], 1.0, torch.Tensor([]),
                             st["viewmatrix"], torch.eye(4).cuda(), H, W, st["beams"], torch.Tensor([]), 1, torch.zeros(3).cuda(), False, 80, 0, False)
cnt = _C.last_counters()
print(cnt)
print("flagged / binned = %.3f" % (cnt["taken_instances"] / max(1, cnt["instances"])))
