"""One-off wider sweep of tests/test_parity_gpu.py::test_random_small_scenes: seeds given on the command line (default 100..159)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_parity_gpu as T
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100, 160)
bad = []
for seed in range(lo, hi):
    try:
        T.test_random_small_scenes(seed, None)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
        print("SEED", seed, "FAILED:", str(e)[:300])
print("failed seeds:", bad)
