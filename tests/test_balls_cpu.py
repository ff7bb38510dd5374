"""Balls (a medium bounded by a sphere that nothing else reaches into): the short cut for rays
that start inside one must change nothing but the work.  Here on the oracle: its results with
the short cut equal its results without it (ORC_NO_BALLS), equal the brute force over all
primitives, and the library finds the same balls."""
import numpy as np

from raytracer_amd import abi, lib
from raytracer_amd.scene import look_at, perspective_camera
from tests import orc
from tests.conftest import random_rays
from tests.mixed import ball_scene, rays_inside


def both(fr, monkeypatch, *args):
    with_balls = orc.OracleScene(fr, *args)
    monkeypatch.setenv("ORC_NO_BALLS", "1")
    without = orc.OracleScene(fr, *args)
    monkeypatch.delenv("ORC_NO_BALLS")
    return with_balls, without


def test_which_balls_are_found(monkeypatch):
    sd, balls = ball_scene(seed=3)
    fr = sd.freeze()
    o, plain = both(fr, monkeypatch)
    assert o.info().n_enclosing == 1
    assert o.info().n_balls == 3 and plain.info().n_balls == 0  # A, B, D; C's neighbour is too close
    g = lib.Scene(fr)
    assert g.info().n_balls == 3 and g.info().n_enclosing == 1
    assert g.nodes().tobytes() == o.nodes().tobytes() == plain.nodes().tobytes()  # the tree does not change


def test_hits_do_not_depend_on_the_short_cut(monkeypatch):
    for seed, max_prims, split in ((3, 4, abi.RT_SPLIT_SAH), (4, 1, abi.RT_SPLIT_HLBVH), (5, 16, abi.RT_SPLIT_MIDDLE)):
        sd, balls = ball_scene(seed=seed)
        o, plain = both(sd.freeze(), monkeypatch, max_prims, split)
        wb = o.world_bound()
        rays = np.concatenate([rays_inside(balls, 6000, seed), random_rays(20000, np.maximum(wb[:3], -45),
                                                                            np.minimum(wb[3:], 45), seed)])
        a, ca = o.intersect(rays, counters=True)
        b, cb = plain.intersect(rays, counters=True)
        assert a.tobytes() == b.tobytes()
        assert (o.occluded(rays) == plain.occluded(rays)).all()
        assert ca[0] < cb[0]  # fewer nodes visited
        kinds = o.frozen._keep["prims"]["kind"] & 0xFF
        hit = a["prim"] != abi.RT_MISS
        assert (kinds[a["prim"][hit]] == abi.RT_MEDIUM).sum() > 1000  # scatters inside the balls


def test_brute_force_agrees(monkeypatch):
    sd, balls = ball_scene(seed=6, n_each=8)
    fr = sd.freeze()
    o = orc.OracleScene(fr)
    rays = rays_inside(balls, 1500, 6)
    got, ref = o.intersect(rays), o.intersect_brute(rays)
    assert (got["prim"] == ref["prim"]).all() and (got["t"] == ref["t"]).all()


def test_frames_do_not_depend_on_the_short_cut(monkeypatch):
    sd, balls = ball_scene(seed=7)
    o, plain = both(sd.freeze(), monkeypatch)
    w, h = 96, 64
    for eye, at in (((30.0, 2.0, 18.0), balls[0][0]), ((-30.0, 6.0, 14.0), balls[1][0]), ((4.0, -30.0, 16.0), balls[3][0])):
        cam = perspective_camera(w, h, 45.0, look_at(eye, at, (0.0, 1.0, 0.0)))
        for flags in (abi.RT_RENDER_COUNTERS, abi.RT_RENDER_COUNTERS | abi.RT_RENDER_DIRECT_LIGHT):
            a, sa = o.render(cam, orc.render_opts(w, h, 4, 40, seed=2, flags=flags))
            b, sb = plain.render(cam, orc.render_opts(w, h, 4, 40, seed=2, flags=flags))
            assert a.tobytes() == b.tobytes()
            assert sa.segments == sb.segments
            assert sa.nodes_visited < sb.nodes_visited
