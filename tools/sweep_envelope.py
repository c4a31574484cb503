"""HIP-vs-oracle residue of sweep scenes against the band of the reference itself.

    python tools/sweep_envelope.py 960017 960752 ...      (GPU)

For every seed (a 3-D scene of tools/parity_sweep.py, lidargs_scenes.sweep_case): the HIP path and the oracle, then the oracle again
under lgo_set_ulp_perturbation (every cos / sin / atan2 / tan / exp result moved inside its CUDA-libdevice error bound: pseudo-random
with two seeds, all up, all down).  Per gradient array: the entries where HIP is off by more than 1e-4 (tests/util.py's metric), how many
of them the perturbed oracles also move by more than 1e-4, and how far HIP lies outside the envelope [min, max] of the five oracle
runs, in units of the envelope's own width at that entry (tests/util.py oracle_envelope / envelope_residue; the regression tests
built from this are tests/test_sweep_residue_gpu.py)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lidargs_scenes as sc
from util import GRAD_KEYS_SR, envelope_residue, hip_forward_backward, oracle_envelope

if __name__ == "__main__":
    for seed in (int(a) for a in sys.argv[1:]):
        scene, W, H, grads, kw, desc = sc.sweep_case(seed, mid=False)
        hip = hip_forward_backward(scene, W, H, grads, **kw)
        base, lo, hi = oracle_envelope(scene, W, H, grads, kw, GRAD_KEYS_SR)
        desc["radii_mismatches"] = int((hip["radii"] != base["radii"]).sum())
        desc["arrays"] = {k: envelope_residue(hip, base, lo, hi, k) for k in GRAD_KEYS_SR}
        print(json.dumps(desc)); sys.stdout.flush()
