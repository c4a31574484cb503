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
