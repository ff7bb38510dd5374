"""Media that enclose the scene, on the GPU: kept out of the tree and tested by the ray's
producer, or (RT355_NO_ENCLOSING) in the tree like every other primitive as BVH::new has them
(shapes/src/bvh.rs:108-151).  Both ways equal the oracle built the same way bit for bit, and
hits, any-hit answers, frames and segment counts equal one another."""
import numpy as np
import pytest

from raytracer_amd import abi, lib
from raytracer_amd.scene import look_at, perspective_camera
from tests import orc
from tests.conftest import random_rays, set_node_source
from tests.mixed import ball_scene, mixed_scene, rays_inside
from tests.test_parity_gpu import assert_hits_equal

pytestmark = pytest.mark.gpu


def both(fr, monkeypatch, *args):
    kept_out = lib.Scene(fr, *args)
    monkeypatch.setenv("RT355_NO_ENCLOSING", "1")
    monkeypatch.setenv("ORC_NO_ENCLOSING", "1")
    literal, oracle = lib.Scene(fr, *args), orc.OracleScene(fr, *args)
    monkeypatch.delenv("RT355_NO_ENCLOSING")
    monkeypatch.delenv("ORC_NO_ENCLOSING")
    return kept_out, literal, oracle


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
@pytest.mark.parametrize("seed,max_prims,split", [(11, 4, abi.RT_SPLIT_SAH), (12, 1, abi.RT_SPLIT_HLBVH)])
def test_hits(monkeypatch, nodes, seed, max_prims, split):
    set_node_source(monkeypatch, nodes)
    sd, balls = ball_scene(seed=seed)
    g, literal, o = both(sd.freeze(), monkeypatch, max_prims, split)
    assert g.info().n_enclosing == 1 and literal.info().n_enclosing == 0 and o.info().n_enclosing == 0
    assert literal.nodes().tobytes() == o.nodes().tobytes()
    wb = o.world_bound()
    rays = np.concatenate([rays_inside(balls, 3000, seed),
                           random_rays(30000, np.maximum(wb[:3], -45), np.minimum(wb[3:], 45), seed)])
    ref = o.intersect(rays)
    assert_hits_equal(literal.intersect(rays), ref, "media in the tree")
    assert_hits_equal(g.intersect(rays), ref, "media kept out of the tree")
    occ = o.occluded(rays)
    assert (literal.does_intersect(rays) == occ).all() and (g.does_intersect(rays) == occ).all()


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_frames(monkeypatch, nodes):
    set_node_source(monkeypatch, nodes)
    sd, balls = ball_scene(seed=16)
    g, literal, o = both(sd.freeze(), monkeypatch)
    w, h = 96, 64
    for eye, at in (((30.0, 2.0, 18.0), balls[0][0]), ((4.0, -30.0, 16.0), (0.0, 0.0, 0.0))):
        cam = perspective_camera(w, h, 45.0, look_at(eye, at, (0.0, 1.0, 0.0)))
        for flags in (abi.RT_RENDER_COUNTERS, abi.RT_RENDER_DIRECT_LIGHT):
            ref, rst = o.render(cam, orc.render_opts(w, h, 4, 40, seed=5, flags=flags))
            a, sa = g.render(cam, lib.render_opts(w, h, 4, 40, seed=5, flags=flags))
            b, sb = literal.render(cam, lib.render_opts(w, h, 4, 40, seed=5, flags=flags))
            assert b.tobytes() == ref.tobytes() and a.tobytes() == ref.tobytes(), (eye, flags)
            assert sa.segments == sb.segments == rst.segments
            if flags & abi.RT_RENDER_COUNTERS:  # the literal tree's own counters
                assert sb.nodes_visited == rst.nodes_visited and list(sb.prims_tested) == list(rst.prims_tested)


def test_week_frame(monkeypatch, scene_cache):
    name, cam, fr = scene_cache("week")
    g, literal, o = both(fr, monkeypatch)
    assert g.info().n_enclosing == 1 and literal.info().n_enclosing == 0
    assert literal.nodes().tobytes() == o.nodes().tobytes()
    w, h = 160, 96
    opts = lib.render_opts(w, h, 8, 50, seed=4)
    a, sa = g.render(cam(w, h), opts)
    b, sb = literal.render(cam(w, h), opts)
    assert a.tobytes() == b.tobytes()
    assert sa.segments == sb.segments and list(sa.bounce_rays) == list(sb.bounce_rays)
