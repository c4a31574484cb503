"""Lane utilisation of the walks on a bench workload: per walked entry, the pixels (of the wave's 64) that take it and the 16-pixel rows
(of 4) that hold one.  Needs an instrumented build:
    LIDARGS_EXTRA_HIPCC_FLAGS=-DLG_LANE_STATS python lidar-gs_amd/build_hip.py --force && python tools/lane_stats.py cfg3 ; python lidar-gs_amd/build_hip.py --force
"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import lidargs_scenes as sc
from diff_lidargs_rasterization import _C
_C.counters_enable(True)      # diagnostics tool: every forward ends with the counting launches of last_counters()
from util import hip_forward_backward

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
kind, P, H, W, seed = {"cfg3": ("street", 2000000, 64, 2650, 3), "cfg2": ("street", 500000, 64, 2650, 2), "cfg4": ("shell", 8000000, 128, 4096, 4)}[wl]
scene = sc.make_scene(kind, P, H, seed)
grads = sc.upstream_grads(H, W, seed)
lib = _C._lib
buf = (C.c_ulonglong * 16)()
lib.lidargs_debug_lane_stats(buf, C.c_int(1))
hip_forward_backward(scene, W, H, grads)
torch.cuda.synchronize()
lib.lidargs_debug_lane_stats(buf, C.c_int(0))
v = list(buf)
def grp(b, name):
    n = max(1, v[b]); nz = max(1, v[b + 3])
    return {"walk": name, "entries_evaluated": v[b], "entries_with_a_taker": v[b + 3], "mean_taking_lanes_per_entry": round(v[b + 1] / n, 2),
            "mean_taking_lanes_per_entry_with_a_taker": round(v[b + 1] / nz, 2), "mean_rows_with_a_taker_per_entry_with_a_taker": round(v[b + 2] / nz, 2)}
print(json.dumps({"workload": wl, "counters": _C.last_counters(), "walks": [grp(0, "T-only walk (pass 1), hit lanes"), grp(4, "full walk (pass 2), blending lanes"), grp(8, "backward walk, contributing lanes")]}, indent=1))
