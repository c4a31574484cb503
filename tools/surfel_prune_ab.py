"""A surfel scene of tools/parity_sweep.py by seed, rendered with and without the footprint pruning (LIDARGS_NO_PRUNE=1 in a second process) and
against the oracle:  python tools/surfel_prune_ab.py <seed>.  What is printed: instances binned either way, entries of every output that differ
between the two HIP runs by more than 2e-5 (the summation grouping moves with the list lengths), and each run's entries over 1e-4 / 1e-3 of the oracle."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
KEYS = ("color", "others", "dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations")
if len(sys.argv) > 2:      # child: render, save
    import lidargs_scenes as sc
    from util import hip_surfel_forward_backward
    from diff_lidargs_rasterization import _C
    c = sc.sweep_case_any(int(sys.argv[1]))
    hip = hip_surfel_forward_backward(c["scene"], c["W"], c["H"], c["grads"], **c["kw"])
    np.savez(sys.argv[2], instances=_C.last_counters()["instances"], radii=hip["radii"], **{k: hip[k] for k in KEYS})
    sys.exit(0)
seed = int(sys.argv[1])
import lidargs_scenes as sc
from util import oracle_surfel_forward_backward
c = sc.sweep_case_any(seed)
assert c["surfel"], "not a surfel seed (seed % 4 == 3 and H >= 4)"
print(c["desc"])
ref = oracle_surfel_forward_backward(c["scene"], c["W"], c["H"], c["grads"], **c["kw"])
res = {}
with tempfile.TemporaryDirectory() as tmp:
    for name, env in (("pruned", {}), ("unpruned", {"LIDARGS_NO_PRUNE": "1"})):
        out = os.path.join(tmp, name + ".npz")
        subprocess.run([sys.executable, __file__, str(seed), out], env=dict(os.environ, **env), check=True)
        res[name] = dict(np.load(out))
a, b = res["pruned"], res["unpruned"]
print("instances binned:", int(a["instances"]), "pruned,", int(b["instances"]), "unpruned; radii equal:", bool(np.array_equal(a["radii"], b["radii"])))
for k in KEYS:
    r = ref[k].astype(np.float64)
    den = np.abs(r) + 1e-3 * np.abs(r).max() + 1e-30
    ab = np.abs(a[k].astype(np.float64) - b[k]) / den
    ea, eb = np.abs(a[k] - r) / den, np.abs(b[k] - r) / den
    print(f"{k:14s} n={r.size:9d}  pruned vs unpruned > 2e-5: {int((ab > 2e-5).sum()):5d} (max {ab.max():.2e})   vs oracle soft/flips: pruned {int(((ea > 1e-4) & (ea <= 1e-3)).sum())}/{int((ea > 1e-3).sum())}"
          f"  unpruned {int(((eb > 1e-4) & (eb <= 1e-3)).sum())}/{int((eb > 1e-3).sum())}")
