"""RT_RENDER_DIRECT_LIGHT on the GPU: the shadow rays Interaction::spawn_ray_to makes inside the
bounce kernel go through the any-hit engine (k_shadow) between bounces.  Images, ray segment
counts and occlusion-ray counts are compared with the oracle bit for bit, for trees read from
LDS and from global memory, several sample batches and tile shards."""
import numpy as np
import pytest

from raytracer_amd import abi, lib
from tests import orc
from tests.mixed import mixed_scene
from tests.test_direct_light_cpu import floor_under_light, sphere_and_rect_lights
from tests.conftest import set_node_source

pytestmark = pytest.mark.gpu
FLAG = abi.RT_RENDER_DIRECT_LIGHT


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
@pytest.mark.parametrize("name,w,h,spp,depth", [("cornell", 96, 96, 4, 50), ("week", 128, 96, 4, 50)])
def test_direct_light_bit_exact(monkeypatch, scene_cache, nodes, name, w, h, spp, depth):
    set_node_source(monkeypatch, nodes)
    _, cam, fr = scene_cache(name)
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    c = cam(w, h)
    ref, rst = o.render(c, orc.render_opts(w, h, spp, depth, seed=2, flags=FLAG | abi.RT_RENDER_COUNTERS))
    got, gst = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, flags=FLAG | abi.RT_RENDER_COUNTERS))
    assert rst.occlusion_rays > w * h  # the any-hit engine really is part of the frame
    assert gst.occlusion_rays == rst.occlusion_rays and gst.segments == rst.segments
    assert gst.nodes_visited == rst.nodes_visited and list(gst.prims_tested) == list(rst.prims_tested)
    assert got.tobytes() == ref.tobytes(), "max abs diff %g" % np.abs(got - ref).max()
    assert gst.trace_launches == depth  # one launch per bounce (plus the shadow launches)
    # several batches of samples, and tile shards that add up to the frame
    got2, st2 = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, flags=FLAG, max_paths=w * h))
    assert st2.batches > 1 and got2.tobytes() == ref.tobytes() and st2.occlusion_rays == rst.occlusion_rays
    total = np.zeros_like(got)
    for k in range(3):
        part, _ = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, flags=FLAG, shard=(k, 3)))
        total += part
    assert total.tobytes() == ref.tobytes()
    # the flag really changes the estimator
    plain, pst = g.render(c, lib.render_opts(w, h, spp, depth, seed=2))
    assert pst.occlusion_rays == 0 and plain.tobytes() != got.tobytes()


def test_direct_light_with_several_lights_and_every_primitive_kind():
    s = mixed_scene(seed=6, n_each=30)
    light = s.diffuse_light((6.0, 5.0, 4.0))
    s.add_rect(1, 14.0, -6.0, 6.0, -6.0, 6.0, light)
    s.add_rect(0, -14.0, -3.0, 3.0, -3.0, 3.0, s.diffuse_light((1.0, 2.0, 8.0)), reverse=True)
    s.add_rect(2, 13.0, -2.0, 2.0, -2.0, 2.0, light)
    fr = s.freeze()
    from raytracer_amd.scene import look_at, perspective_camera
    w, h, spp = 100, 70, 3
    c = perspective_camera(w, h, 50.0, look_at((0, 3, -30), (0, 0, 0), (0, 1, 0)), lens_radius=0.2,
                           focal_distance=30.0, shutter=(0.0, 1.0))
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    ref, rst = o.render(c, orc.render_opts(w, h, spp, 12, seed=7, flags=FLAG))
    got, gst = g.render(c, lib.render_opts(w, h, spp, 12, seed=7, flags=FLAG))
    assert rst.occlusion_rays > 1000 and gst.occlusion_rays == rst.occlusion_rays and gst.segments == rst.segments
    assert got.tobytes() == ref.tobytes()


def test_direct_light_against_the_closed_form_on_the_gpu():
    """The GPU's own image against float64 quadrature of the light's form factor (no oracle in
    the comparison): a Lambertian floor under a rectangular light, one-segment paths."""
    albedo, emit = 0.6, 5.0
    s, cam = floor_under_light(albedo, emit)
    g = lib.Scene(s.freeze())
    w = h = 32
    img, st = g.render(cam, lib.render_opts(w, h, 1024, 1, seed=4, flags=FLAG))
    assert st.occlusion_rays > 0
    rays = lib.generate_rays(cam, lib.render_opts(w, h, 1, sampler=abi.RT_SAMPLER_STRATIFIED,
                                                  sampler_arg=1 | abi.RT_STRATIFIED_NO_JITTER), 0, w * h)
    hits = g.intersect(rays)
    on_floor = hits["prim"] == 0
    P = rays["o"].astype(np.float64) + hits["t"][:, None].astype(np.float64) * rays["d"].astype(np.float64)
    gx, gz = np.meshgrid(np.linspace(-1, 1, 64, endpoint=False) + 1 / 64, np.linspace(-0.5, 0.5, 32, endpoint=False) + 1 / 64)
    Q = np.stack([gx.ravel(), np.full(gx.size, 3.0), gz.ravel()], axis=1)
    d = Q[None, :, :] - P[:, None, :]
    d2 = (d * d).sum(-1)
    E = ((d[..., 1] / np.sqrt(d2)) ** 2 / d2).mean(axis=1) * 2.0
    want = albedo / np.pi * emit * E
    got = img[..., 0].reshape(-1)
    m = on_floor & (want > 0.02 * want.max())
    assert abs(got[m].sum() / want[m].sum() - 1.0) < 0.005
    assert np.abs(got[m] / want[m] - 1.0).max() < 0.12


def test_shadow_queue_keeps_its_own_capacity():
    """A large plain frame grows the bounce queues, a small direct-light frame then sizes the
    shadow queue: the two capacities differ, and the shadow rays must still land in (and be read
    from) their own sub-queues.  Both orders, on one Scene."""
    _, cam, fr = scene_cache_free("cornell")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    big, _ = g.render(cam(256, 192), lib.render_opts(256, 192, 4, 6, seed=1))
    w, h, spp = 64, 48, 2
    c = cam(w, h)
    ref, rst = o.render(c, orc.render_opts(w, h, spp, 8, seed=3, flags=FLAG))
    got, gst = g.render(c, lib.render_opts(w, h, spp, 8, seed=3, flags=FLAG))
    assert gst.occlusion_rays == rst.occlusion_rays > 0 and got.tobytes() == ref.tobytes()
    # direct small, plain larger still, direct small again
    g.render(cam(320, 256), lib.render_opts(320, 256, 4, 6, seed=1))
    got, gst = g.render(c, lib.render_opts(w, h, spp, 8, seed=3, flags=FLAG))
    assert gst.occlusion_rays == rst.occlusion_rays and got.tobytes() == ref.tobytes()


def scene_cache_free(name):
    from raytracer_amd import scenes
    sd, cam = scenes.BUILDERS[name]()
    return sd, cam, sd.freeze()


def test_emissive_sphere_beside_a_sampled_light():
    s, cam = sphere_and_rect_lights()
    fr = s.freeze()
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w = h = 24
    ref, rst = o.render(cam, orc.render_opts(w, h, 16, 4, seed=5, flags=FLAG))
    got, gst = g.render(cam, lib.render_opts(w, h, 16, 4, seed=5, flags=FLAG))
    assert gst.occlusion_rays == rst.occlusion_rays > 0 and gst.segments == rst.segments
    assert got.tobytes() == ref.tobytes()


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_distant_light_bit_exact(monkeypatch, nodes):
    """LightSource "distant": alone (closed-form floor, see the CPU test), sharing the choice with
    a lamp, and on test.pbrt's scene with several sample batches and shards."""
    from tests.test_direct_light_cpu import floor_under_sun
    set_node_source(monkeypatch, nodes)
    w = h = 32
    for kw in (dict(), dict(lamp=True), dict(ball=False, towards=(1.0, -2.0, -0.5))):
        s, cam = floor_under_sun(**kw)
        fr = s.freeze()
        g, o = lib.Scene(fr), orc.OracleScene(fr)
        for depth in (1, 6):
            ref, rst = o.render(cam, orc.render_opts(w, h, 4, depth, seed=3, flags=FLAG))
            got, gst = g.render(cam, lib.render_opts(w, h, 4, depth, seed=3, flags=FLAG))
            assert got.tobytes() == ref.tobytes(), (kw, depth)
            assert gst.occlusion_rays == rst.occlusion_rays and gst.segments == rst.segments


def test_distant_light_on_the_pbrt_scene(scene_cache):
    _, cam, fr = scene_cache("pbrt")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp, depth = 120, 80, 4, 20
    c = cam(w, h)
    kw = dict(sampler=abi.RT_SAMPLER_HALTON)
    ref, rst = o.render(c, orc.render_opts(w, h, spp, depth, seed=2, flags=FLAG | abi.RT_RENDER_COUNTERS, **kw))
    got, gst = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, flags=FLAG | abi.RT_RENDER_COUNTERS, **kw))
    assert rst.occlusion_rays > w * h
    assert gst.occlusion_rays == rst.occlusion_rays and gst.segments == rst.segments
    assert gst.nodes_visited == rst.nodes_visited and list(gst.prims_tested) == list(rst.prims_tested)
    assert got.tobytes() == ref.tobytes()
    got2, st2 = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, flags=FLAG, max_paths=w * h, **kw))
    assert st2.batches > 1 and got2.tobytes() == ref.tobytes()
    total = np.zeros_like(got)
    for k in range(3):
        part, _ = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, flags=FLAG, shard=(k, 3), **kw))
        total += part
    assert total.tobytes() == ref.tobytes()
    plain, pst = g.render(c, lib.render_opts(w, h, spp, depth, seed=2, **kw))
    assert pst.occlusion_rays == 0 and plain.tobytes() != got.tobytes()
