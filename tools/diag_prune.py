"""Diagnose image differences between the HIP path and the oracle on the pruning stress scene of tests/test_beam_tables_gpu.py:
runs it under several internal-knob settings in sub-processes (pruning off, forced tile heights, segment plans) and prints the
outlier counts per setting and, for the default setting, the worst pixels.   python tools/diag_prune.py [table]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r"""
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import lidargs_scenes as sc
from util import hip_forward_backward
from test_beam_tables_gpu import _stress_scene
from diff_lidargs_rasterization import _C
_C.counters_enable(True)      # diagnostics tool: every forward ends with the counting launches of last_counters()
scene, W, H, grads = _stress_scene(%r)
hip = hip_forward_backward(scene, W, H, None)
c = _C.last_counters()
np.savez(%r, color=hip["color"], depth=hip["depth"], occ=hip["occ"], radii=hip["radii"], cnt=np.array([c["tile_rows"], c["segments"], c["instances"]]))
"""
def run(table, env, out):
    code = CODE % (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests"), table, out)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    if r.returncode:
        print(r.stdout[-2000:], r.stderr[-3000:]); raise SystemExit(1)
if __name__ == "__main__":
    import numpy as np
    sys.path[:0] = [ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")]
    table = sys.argv[1] if len(sys.argv) > 1 else "waymo"
    from test_beam_tables_gpu import _stress_scene
    from util import oracle_forward_backward
    scene, W, H, grads = _stress_scene(table)
    ref = oracle_forward_backward(scene, W, H, None)
    variants = [("default", {}), ("no_prune", {"LIDARGS_NO_PRUNE": "1"}), ("rows4", {"LIDARGS_TILE_ROWS": "4"}), ("rows8", {"LIDARGS_TILE_ROWS": "8"}),
                ("rows16", {"LIDARGS_TILE_ROWS": "16"}), ("rows32", {"LIDARGS_TILE_ROWS": "32"}), ("seg128_nohead", {"LIDARGS_SEG_LEN": "128", "LIDARGS_HEAD": "0"}),
                ("head1", {"LIDARGS_HEAD": "1"}), ("one_segment", {"LIDARGS_MAX_SEGMENTS": "1"}), ("norounds", {"LIDARGS_ROUNDS": ""})]
    keep = None
    for name, env in variants:
        out = f"/tmp/diag_{name}.npz"
        run(table, env, out)
        h = np.load(out)
        if name == "default":
            keep = h
        line = f"{name:14s} tile_rows {int(h['cnt'][0]):2d} slots {int(h['cnt'][1]):2d} instances {int(h['cnt'][2]):8d} |"
        for k in ("color", "depth", "occ"):
            r = ref[k]; d = np.abs(h[k] - r) / (np.abs(r) + 1e-3 * np.abs(r).max())
            line += f" {k} >1e-4: {(d > 1e-4).sum():4d} max {d.max():.2e} |"
        print(line)
    a = keep
    d = np.abs(a["color"] - ref["color"]).max(0)
    ys, xs = np.nonzero(d > 1e-3)
    print("bad pixels (default vs oracle):", len(ys), "rows:", sorted(set(ys.tolist())))
    f = ref["fwd"]
    ranges = f.array("ranges").reshape(-1, 2); nc = f.array("n_contrib").reshape(H, W)
    tiles_x = (W + 15) // 16
    for y, x in list(zip(ys, xs))[:16]:
        t = y * tiles_x + x // 16
        print(f" pixel (y={y}, x={x}): hip C0 {a['color'][0, y, x]:.5f} C1 {a['color'][1, y, x]:.5f} D {a['depth'][0, y, x]:.4f} occ {a['occ'][0, y, x]:.6f} | oracle C0 {ref['color'][0, y, x]:.5f} "
              f"C1 {ref['color'][1, y, x]:.5f} D {ref['depth'][0, y, x]:.4f} occ {ref['occ'][0, y, x]:.6f} | oracle tile list {ranges[t, 1] - ranges[t, 0]} entries, n_contrib {nc[y, x]}")
