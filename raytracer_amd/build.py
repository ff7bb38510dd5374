"""Builds raytracer_amd/librt355.so for gfx950 with hipcc (in-tree, no JIT cache).

-ffp-contract=off is part of the numerical contract: every multiply-add must
round twice so that hit/miss decisions match the reference arithmetic.

Each source is compiled to its own object (in parallel, only when it or a header
is newer than the object) and the objects are linked into the shared library.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "librt355.so")
SOURCES = ["rt_kernels.hip", "rt_build_gpu.hip", "rt_api.cpp", "rt_bvh.cpp"]
HEADERS = ["rt_device.hpp", "rt_kernels.hpp", "rt_bvh.hpp", "rt_probe.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-pthread", "-Wall", "-Wno-unused-function"]


def _deps():
    deps = [os.path.join(CSRC, f) for f in HEADERS]
    deps.append(os.path.join(ROOT, "include", "rt355.h"))
    return deps


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps() + [os.path.join(CSRC, f) for f in SOURCES])


def build(force=False, verbose=False, extra=(), lib=LIB):
    """Compiles what is out of date and links `lib`.  `extra` flags (e.g. -D probes) get
    their own object directory, so variant libraries do not disturb the shipped one."""
    if not force and not extra and not stale(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tag = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8] if extra else "base"
    objdir = os.path.join(OBJ, tag)
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(d) for d in _deps())

    def compile_one(src):
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(path)
                and os.path.getmtime(obj) > newest_header):
            return obj
        cmd = [hipcc, *FLAGS, *extra, "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-x", "hip", "-c", path,
               "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else ())
