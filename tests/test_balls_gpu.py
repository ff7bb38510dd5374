"""Balls on the GPU: rays that start inside a fog ball are resolved by the ball's own primitives
(by whoever makes the ray; for rays from the caller by the traversal's first turn), and shading
batches of the one-launch kernel go round again in place for them.  Everything must equal the
oracle bit for bit -- hits, any-hit, counters, frames, statistics -- and must not depend on the
short cut being there (RT355_NO_BALLS)."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, lib
from raytracer_amd.scene import look_at, perspective_camera
from tests import orc
from tests.conftest import random_rays, scene_with_hooks, set_node_source
from tests.mixed import ball_scene, rays_inside
from tests.test_parity_gpu import assert_hits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
@pytest.mark.parametrize("seed,max_prims,split", [(3, 4, abi.RT_SPLIT_SAH), (4, 1, abi.RT_SPLIT_HLBVH)] +
                         # a longer soak on request: RT355_SWEEP_EXTRA=<n> more seeds
                         [(s, 1 + s % 7, s % 4) for s in range(100, 100 + int(os.environ.get("RT355_SWEEP_EXTRA", "0")))])
def test_rays_from_inside_the_balls(monkeypatch, nodes, seed, max_prims, split):
    import torch
    set_node_source(monkeypatch, nodes)
    sd, balls = ball_scene(seed=seed)
    fr = sd.freeze()
    g, o = lib.Scene(fr, max_prims, split), orc.OracleScene(fr, max_prims, split)
    assert g.info().n_balls == o.info().n_balls == 3
    wb = o.world_bound()
    rays = np.concatenate([rays_inside(balls, 6000, seed),
                           random_rays(20000, np.maximum(wb[:3], -45), np.minimum(wb[3:], 45), seed)])
    ref, ref_counters = o.intersect(rays, counters=True)
    assert_hits_equal(g.intersect(rays), ref, "balls")
    assert (g.does_intersect(rays) == o.occluded(rays)).all()
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((len(rays), 8), dtype=torch.float32, device="cuda")
    counters = g.intersect_count(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
    assert (counters == ref_counters).all(), (counters, ref_counters)
    assert_hits_equal(d_hits.cpu().numpy().view(abi.hit_dtype).reshape(-1), ref, "balls (counting)")


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
@pytest.mark.parametrize("loop_from", [None, 0, 2])
def test_frames_through_the_balls(monkeypatch, nodes, loop_from):
    set_node_source(monkeypatch, nodes)
    sd, balls = ball_scene(seed=7)
    fr = sd.freeze()
    g = scene_with_hooks(monkeypatch, fr, **({} if loop_from is None else {"RT355_LOOP_FROM": loop_from}))
    o = orc.OracleScene(fr)
    w, h = 96, 64
    for eye, at in (((30.0, 2.0, 18.0), balls[0][0]), ((-30.0, 6.0, 14.0), balls[1][0]), ((4.0, -30.0, 16.0), balls[3][0])):
        cam = perspective_camera(w, h, 45.0, look_at(eye, at, (0.0, 1.0, 0.0)))
        for flags in (0, abi.RT_RENDER_COUNTERS, abi.RT_RENDER_COUNTERS | abi.RT_RENDER_DIRECT_LIGHT):
            if (flags & abi.RT_RENDER_DIRECT_LIGHT) and loop_from is not None:
                continue  # direct lighting keeps one launch per bounce
            img, st = g.render(cam, lib.render_opts(w, h, 4, 40, seed=2, flags=flags))
            ref, rst = o.render(cam, orc.render_opts(w, h, 4, 40, seed=2, flags=flags))
            assert img.tobytes() == ref.tobytes(), (eye, flags)
            assert st.segments == rst.segments
            if flags & abi.RT_RENDER_COUNTERS:
                assert st.nodes_visited == rst.nodes_visited
                assert list(st.prims_tested) == list(rst.prims_tested)


def test_frames_do_not_depend_on_the_short_cut(monkeypatch, scene_cache):
    name, cam, fr = scene_cache("week")
    with_balls = lib.Scene(fr)
    monkeypatch.setenv("RT355_NO_BALLS", "1")
    without = lib.Scene(fr)
    monkeypatch.delenv("RT355_NO_BALLS")
    assert with_balls.info().n_balls == 1 and without.info().n_balls == 0
    w, h = 160, 96
    opts = lib.render_opts(w, h, 8, 50, seed=4, flags=abi.RT_RENDER_COUNTERS)
    a, sa = with_balls.render(cam(w, h), opts)
    b, sb = without.render(cam(w, h), opts)
    assert a.tobytes() == b.tobytes()
    assert sa.segments == sb.segments and list(sa.bounce_rays) == list(sb.bounce_rays)
    assert sa.nodes_visited < sb.nodes_visited
