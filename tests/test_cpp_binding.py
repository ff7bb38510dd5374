"""The C++ mirror of the reference's aggregate interface (bindings/cpp/rt355.hpp)
compiles against include/rt355.h, links librt355.so and behaves: on a GPU it
returns the known hits of three rays; without one the compute calls fail with
RT_ERR_NO_DEVICE (never a CPU answer)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(tmp_path):
    exe = str(tmp_path / "example")
    libdir = os.path.join(ROOT, "raytracer_amd")
    assert os.path.exists(os.path.join(libdir, "librt355.so")), "build librt355.so first (python -m raytracer_amd.build)"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "bindings", "cpp"),
                           os.path.join(ROOT, "bindings", "cpp", "example.cpp"),
                           "-L", libdir, "-lrt355", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_cpp_mirror_builds_and_reports_no_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    out = subprocess.run([build_example(tmp_path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "world_bound [-10 -1 -20] [10 1 0] nodes 5" in out.stdout
    assert "no device" in out.stdout and "no CPU path" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_hits(tmp_path):
    out = subprocess.run([build_example(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok"), out.stdout
