import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def random_rays(n, lo, hi, seed, finite_fraction=0.25, axis_fraction=0.05):
    """Incoherent rays: origins uniform in the box [lo, hi] (slightly enlarged),
    directions uniform on the sphere; a share has a finite t_max and a share is
    axis aligned (zero direction components exercise the 1/0 slab path)."""
    from raytracer_amd import abi
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    ext = hi - lo
    rays = np.zeros(n, dtype=abi.ray_dtype)
    rays["o"] = (lo - 0.1 * ext + rng.random((n, 3)) * 1.2 * ext).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = int(n * axis_fraction)
    if k:
        d[:k] = 0.0
        d[np.arange(k), rng.integers(0, 3, k)] = rng.choice([-1.0, 1.0], k)
    rays["d"] = d.astype(np.float32)
    rays["t_max"] = np.inf
    m = int(n * finite_fraction)
    if m:
        idx = rng.choice(n, m, replace=False)
        rays["t_max"][idx] = (rng.random(m) * np.linalg.norm(ext)).astype(np.float32)
    rays["time"] = rng.random(n).astype(np.float32)
    return rays


@pytest.fixture(scope="session")
def scene_cache():
    """name -> (SceneData, camera factory, frozen description); built once."""
    from raytracer_amd import scenes
    cache = {}

    def get(name, **kw):
        key = (name, tuple(sorted(kw.items())))
        if key not in cache:
            sd, cam = scenes.BUILDERS[name](**kw)
            cache[key] = (sd, cam, sd.freeze())
        return cache[key]

    return get


def scene_with_hooks(monkeypatch, frozen, *args, **env):
    """A library scene built and placed on the device while the given RT355_* test hooks are set
    (the library reads them once, at that moment; later changes of the environment do nothing)."""
    from raytracer_amd import lib
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    g = lib.Scene(frozen, *args)
    g.upload()
    for k in env:
        monkeypatch.delenv(k)
    return g


def set_node_source(monkeypatch, nodes):
    """Test parameter -> library hooks, read when the scene is placed on the device.  "lds": the
    library's own choice (small trees live in LDS); "global": two-box records in global memory, and
    four-box records from the third bounce of a frame on, as for large scenes; "quads": global memory
    with four-box records in every launch that has them (rt_intersect, rt_occluded, all bounces)."""
    if nodes in ("global", "quads"):
        monkeypatch.setenv("RT355_NODES", "global")
    if nodes == "quads":
        monkeypatch.setenv("RT355_QUADS", "2")
