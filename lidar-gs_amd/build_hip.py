"""Compile the gfx950 rasterizer into an in-tree shared library (no torch involved).

    python lidar-gs_amd/build_hip.py [--force]

Output: lidar-gs_amd/diff_lidargs_rasterization/liblidargs_hip.so  (git-ignored, travels with gpurun)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "diff_lidargs_rasterization", "liblidargs_hip.so")
SOURCES = ["api.hip", "preprocess.hip", "binning.hip", "render.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-fgpu-rdc" if False else "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "lidargs_rasterizer.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
