import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "lidar-gs_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a HIP device: without one they are SKIPPED (plain `pytest tests` stays green on a GPU-less box);
    on the GPU box nothing is skipped, so a missing device there cannot hide behind a green run (the driver's -m gpu tier
    checks that the native library was really loaded)."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib_built():
    """Build the HIP library if needed (hipcc cross-compiles without a GPU)."""
    import build_hip
    return build_hip.build()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """How much of the parity budget of tests/util.py the session really used: worst case per quantity over every parity() call
    (HIP vs oracle).  Printed, and written to gpurun_out/parity_budget.json on the GPU box so that the numbers quoted in DESIGN.md
    section 3 come from a file."""
    try:
        import util
    except Exception:
        return
    log = [s for s in util.PARITY_LOG if s.get("n", 0) > 0]
    if not log:
        return
    worst_p999 = max(log, key=lambda s: s["p999"])
    worst_max = max(log, key=lambda s: s["max"])
    worst_frac = max(log, key=lambda s: s["soft_frac_used"])
    worst_flip = max(log, key=lambda s: s["flip_frac_used"])
    worst_abs = max(log, key=lambda s: s.get("abs_max", 0.0))
    with_out = [s for s in log if s["outliers"] > 0]
    tr = terminalreporter
    tr.write_sep("-", "parity budget used (tests/util.py: rtol 1e-4 with floor 1e-3 max|ref|)")
    tr.write_line(f"parity() calls: {len(log)}; calls with any entry over rtol: {len(with_out)}")
    tr.write_line(f"worst p99.9 : {worst_p999['p999']:.3e}  ({worst_p999['name']}, n={worst_p999['n']})")
    tr.write_line(f"worst max   : {worst_max['max']:.3e}  ({worst_max['name']}, n={worst_max['n']})")
    tr.write_line(f"worst soft fraction (rtol < err <= 1e-3): {worst_frac['soft_frac_used']:.3e} = {worst_frac['soft']} of {worst_frac['n']} "
                  f"({worst_frac['name']}); allowed {worst_frac['allowed']}")
    tr.write_line(f"worst flip fraction (err > 1e-3): {worst_flip['flip_frac_used']:.3e} = {worst_flip['flips']} of {worst_flip['n']} "
                  f"({worst_flip['name']}); allowed {worst_flip['allowed_flips']}")
    tr.write_line(f"largest single difference: {worst_abs.get('abs_max', 0.0):.3e} x max|ref| ({worst_abs['name']}); allowed {util.FLIP_ABS_MAX}")
    try:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out) or os.environ.get("GRAFT_REPO_ROOT"):
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_budget.json"), "w") as f:
                json.dump(dict(calls=len(log), calls_with_outliers=len(with_out), worst_p999=worst_p999, worst_max=worst_max,
                               worst_soft_fraction=worst_frac, worst_flip_fraction=worst_flip, worst_abs_difference=worst_abs,
                               budgets=dict(soft_frac=util.SOFT_FRAC, flip_frac=util.FLIP_FRAC, flip_abs_max=util.FLIP_ABS_MAX, min_count=util.MIN_COUNT),
                               over_rtol=[dict(name=s["name"], n=s["n"], soft=s["soft"], flips=s["flips"], max=s["max"], abs_max=s.get("abs_max", 0.0),
                                               allowed=s["allowed"], allowed_flips=s["allowed_flips"]) for s in with_out]), f, indent=1)
    except Exception as e:      # the summary must never fail a run
        tr.write_line(f"(parity_budget.json not written: {e})")
