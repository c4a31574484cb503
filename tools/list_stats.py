"""Diagnostic: per-patch list length, segments and how many of them the forward walked (reads the opaque image / binning buffers by
their known layout, csrc/lidargs_common.h img_carve / bin_carve).   python tools/list_stats.py [cfg3]"""
import sys, numpy as np, torch
sys.path[:0] = ["/root/repo", "/root/repo/lidar-gs_amd", "/root/repo/tests"]
import lidargs_scenes as sc
from util import to_torch
from diff_lidargs_rasterization import _C
_C.counters_enable(True)      # diagnostics tool: every forward ends with the counting launches of last_counters()
kind, P, H, W, seed = sc.BASELINE_CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
scene = sc.make_scene(kind, P, H, seed); st = to_torch(scene)
n, color, depth, occ, radii, gb, bb, ib = _C.rasterize_gaussians(st["bg"], st["means3D"], st["colors"], st["opacities"], st["scales"], st["rotations"], 1.0, torch.Tensor([]), st["viewmatrix"], torch.eye(4).cuda(), H, W, st["beams"], torch.Tensor([]), 1, torch.zeros(3).cuda(), False, 80, 0, False)
cnt = _C.last_counters(); TH = cnt["tile_rows"]; S = cnt["segments"]; N = H * W
al = lambda x: (x + 127) // 128 * 128
tiles4 = ((W + 15) // 16) * ((H + 3) // 4)
tiles = ((W + 15) // 16) * ((H + TH - 1) // TH); wpt = TH // 4; patches = tiles * wpt
o_rg = al(al(4 * N) + 4 * N)
buf = ib.cpu().numpy()
ranges = buf[o_rg:o_rg + 8 * tiles].view(np.uint32).reshape(tiles, 2)
L = (ranges[:, 1] - ranges[:, 0]).astype(np.int64)
Rp = n & ~3
nn = max(Rp, 1)
blocks = (nn + 4095) // 4096; chunks = (blocks + 1023) // 1024
scratch_words = 256 * blocks + 256 + 256 * chunks + 64
o = 0
for _ in range(4): o = al(o) + 4 * nn
o = al(o) + 4 * scratch_words
o_seg = al(o); o = o_seg + 4 * patches * S * 7 * 64
o_fl = al(o); o = o_fl + wpt * nn + 64
o_alive = al(o)
bbuf = bb.cpu().numpy()
alive = bbuf[o_alive:o_alive + patches].astype(np.int64)
seg_len = 64 if S == 45 else 128
St = np.minimum(S, np.maximum(1, (L + seg_len - 1) // seg_len))
St_p = np.repeat(St, wpt); L_p = np.repeat(L, wpt)
walked = np.where(alive == 255, St_p, np.minimum(alive, St_p))
seglen_p = (L_p + St_p - 1) // St_p
print(cnt)
print("list len: mean %.0f median %.0f p90 %.0f p99 %.0f max %d" % (L.mean(), np.median(L), np.quantile(L, .9), np.quantile(L, .99), L.max()))
print("segments per patch: mean %.1f p90 %d max %d; walked: mean %.1f median %d p90 %d p99 %d max %d; patches walked to the end %d of %d"
      % (St_p.mean(), np.quantile(St_p, .9), St_p.max(), walked.mean(), np.median(walked), np.quantile(walked, .9), np.quantile(walked, .99), walked.max(), int((walked >= St_p).sum()), patches))
ent = walked * seglen_p
print("entries walked per patch: mean %.0f p90 %.0f p99 %.0f max %d sum %.3g (all lists %.3g)" % (ent.mean(), np.quantile(ent, .9), np.quantile(ent, .99), ent.max(), ent.sum(), L_p.sum()))
for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 32), (32, 40), (40, 46)):
    m = (walked > lo) & (walked <= hi)
    print(f"  walked in ({lo},{hi}]: {int(m.sum())} patches, entries {int(ent[m].sum())}, mean segment length {seglen_p[m].mean() if m.any() else 0:.0f}")
