"""Builds raytracer_amd/librt355.so for gfx950 with hipcc (in-tree, no JIT cache).

-ffp-contract=off is part of the numerical contract: every multiply-add must
round twice so that hit/miss decisions match the reference arithmetic.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt355.so")
SOURCES = ["rt_kernels.hip", "rt_build_gpu.hip", "rt_api.cpp", "rt_bvh.cpp"]
HEADERS = ["rt_device.hpp", "rt_kernels.hpp", "rt_bvh.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-pthread", "-Wall", "-Wno-unused-function"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(ROOT, "include", "rt355.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, *extra, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip",
           *[os.path.join(CSRC, f) for f in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ())
