"""Committed golden vectors (tests/golden/*.npz, written by make_golden.py from the
oracle): rays, closest-hit records, any-hit bytes, traversal counters and a small
image per scene.  The CPU test pins the oracle to them; the GPU test checks the
HIP path against the same files without running the oracle."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, lib, scenes
from tests import orc
from tests.mixed import mixed_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["mixed", "random", "cornell", "week", "pbrt"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd, cam = (mixed_scene(seed=1, n_each=40), None) if name == "mixed" else scenes.BUILDERS[name]()
    rays = np.ascontiguousarray(z["rays"]).view(abi.ray_dtype).reshape(-1)
    hits = np.ascontiguousarray(z["hits"]).view(abi.hit_dtype).reshape(-1)
    return z, sd.freeze(), cam, rays, hits


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    z, fr, cam, rays, hits = load(name)
    o = orc.OracleScene(fr)
    assert o.info().n_nodes == z["n_nodes"]
    got, cnt = o.intersect(rays, counters=True)
    assert got.tobytes() == hits.tobytes()
    assert (cnt == z["counters"]).all()
    assert (o.occluded(rays) == z["occluded"]).all()
    if cam is not None:
        img, st = o.render(cam(64, 48), orc.render_opts(64, 48, 2, 20, seed=11))
        assert img.tobytes() == z["image"].tobytes() and st.segments == z["segments"]
        img, st = o.render(cam(64, 48), orc.render_opts(64, 48, 2, 20, seed=11, flags=abi.RT_RENDER_DIRECT_LIGHT))
        assert img.tobytes() == z["image_direct"].tobytes()
        assert st.segments == z["segments_direct"] and st.occlusion_rays == z["occlusion_direct"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(name):
    z, fr, cam, rays, hits = load(name)
    g = lib.Scene(fr)
    assert g.info().n_nodes == z["n_nodes"]
    assert g.intersect(rays).tobytes() == hits.tobytes()
    assert (g.does_intersect(rays) == z["occluded"]).all()
    if cam is not None:
        img, st = g.render(cam(64, 48), lib.render_opts(64, 48, 2, 20, seed=11))
        assert img.tobytes() == z["image"].tobytes() and st.segments == z["segments"]
        img, st = g.render(cam(64, 48), lib.render_opts(64, 48, 2, 20, seed=11, flags=abi.RT_RENDER_DIRECT_LIGHT))
        assert img.tobytes() == z["image_direct"].tobytes()
        assert st.segments == z["segments_direct"] and st.occlusion_rays == z["occlusion_direct"]
