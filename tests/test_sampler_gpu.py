"""Camera samplers (Halton, Sobol', (0,2), stratified) on the GPU against the oracle, bit for bit: generated rays,
rendered frames (several batches), and the film's regenerated sample positions."""
import numpy as np
import pytest

from raytracer_amd import abi, lib
from tests import orc
from tests.test_film_cpu import FILTERS


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(40, 24), (300, 200)])
def test_halton_rays_and_frames(w, h, scene_cache):
    _, cam, fr = scene_cache("random")  # thin lens: all five dimensions matter
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    spp = 3
    opts = lib.render_opts(w, h, spp, 6, seed=9, sampler=abi.RT_SAMPLER_HALTON)
    n = min(w * h * spp, 50000)
    a = lib.generate_rays(cam(w, h), opts, 17, n)
    b = orc.generate_rays(cam(w, h), orc.render_opts(w, h, spp, 6, seed=9, sampler=abi.RT_SAMPLER_HALTON), 17, n)
    assert a.tobytes() == b.tobytes()
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 6, seed=9, sampler=abi.RT_SAMPLER_HALTON,
                                                  max_paths=((w + 31) // 32) * ((h + 31) // 32) * 1024 * 2))
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 6, seed=9, sampler=abi.RT_SAMPLER_HALTON))
    assert st.batches == 2 and img.tobytes() == ref.tobytes() and st.segments == rst.segments


@pytest.mark.gpu
def test_halton_with_film(scene_cache):
    _, cam, fr = scene_cache("cornell")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp, f = 72, 40, 4, FILTERS["gaussian"]
    img, _ = g.render(cam(w, h), lib.render_opts(w, h, spp, 8, seed=1, filter=f, sampler=abi.RT_SAMPLER_HALTON))
    ref, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=1, filter=f, sampler=abi.RT_SAMPLER_HALTON))
    assert img.tobytes() == ref.tobytes()
    with pytest.raises(lib.RtError):
        g.render(cam(w, h), lib.render_opts(w, h, spp, 8, sampler=7))


SAMPLERS = {
    "sobol": dict(sampler=abi.RT_SAMPLER_SOBOL),
    "zero_two": dict(sampler=abi.RT_SAMPLER_ZERO_TWO),
    "stratified": dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=2),
    "stratified_centres": dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=4 | abi.RT_STRATIFIED_NO_JITTER),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SAMPLERS))
@pytest.mark.parametrize("w,h", [(40, 24), (300, 200)])
def test_sampler_rays_and_frames(name, w, h, scene_cache):
    """Sobol', (0,2) and stratified camera samples: generated rays of a path range that starts
    and ends inside a pixel, and frames rendered in two batches of samples, bit for bit."""
    kw = SAMPLERS[name]
    _, cam, fr = scene_cache("random")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    spp = 4
    n = min(w * h * spp - 17, 50000)
    a = lib.generate_rays(cam(w, h), lib.render_opts(w, h, spp, 6, seed=9, **kw), 17, n)
    b = orc.generate_rays(cam(w, h), orc.render_opts(w, h, spp, 6, seed=9, **kw), 17, n)
    assert a.tobytes() == b.tobytes()
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 6, seed=9, **kw,
                                                  max_paths=((w + 31) // 32) * ((h + 31) // 32) * 1024 * 2))
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 6, seed=9, **kw))
    assert st.batches == 2 and img.tobytes() == ref.tobytes() and st.segments == rst.segments


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sobol", "zero_two", "stratified"])
def test_sampler_with_film_and_shards(name, scene_cache):
    """The film regenerates the sample positions of neighbouring pixels; shards own disjoint
    tiles whose raw films add up to the whole frame's."""
    kw = SAMPLERS[name]
    _, cam, fr = scene_cache("cornell")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp, f = 72, 40, 4, FILTERS["gaussian"]
    img, _ = g.render(cam(w, h), lib.render_opts(w, h, spp, 8, seed=1, filter=f, **kw))
    ref, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=1, filter=f, **kw))
    assert img.tobytes() == ref.tobytes()
    parts = [g.render(cam(w, h), lib.render_opts(w, h, spp, 8, seed=1, shard=(k, 2), **kw))[0] for k in range(2)]
    whole, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=1, **kw))
    for k in range(2):
        part_ref, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=1, shard=(k, 2), **kw))
        assert parts[k].tobytes() == part_ref.tobytes()
    assert (parts[0][..., :3] + parts[1][..., :3]).tobytes() == whole[..., :3].tobytes()


@pytest.mark.gpu
def test_sampler_arguments(scene_cache):
    _, cam, fr = scene_cache("cornell")
    g = lib.Scene(fr)
    for spp, kw in ((6, dict(sampler=abi.RT_SAMPLER_ZERO_TWO)),
                    (6, dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=4)),
                    (6, dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=0)),
                    (4, dict(sampler=5))):
        with pytest.raises(lib.RtError):
            g.render(cam(16, 8), lib.render_opts(16, 8, spp, 4, **kw))
        with pytest.raises(lib.RtError):
            lib.generate_rays(cam(16, 8), lib.render_opts(16, 8, spp, 4, **kw), 0, 8)


@pytest.mark.gpu
def test_sobol_at_frame_size(scene_cache):
    """1920x1080 frame (resolution 2048, m = 11): rays of scattered path ranges."""
    _, cam, fr = scene_cache("random")
    w, h, spp = 1920, 1080, 8
    kw = SAMPLERS["sobol"]
    for first in (0, 5_000_003, w * h * spp - 4096):
        a = lib.generate_rays(cam(w, h), lib.render_opts(w, h, spp, 6, seed=3, **kw), first, 4096)
        b = orc.generate_rays(cam(w, h), orc.render_opts(w, h, spp, 6, seed=3, **kw), first, 4096)
        assert a.tobytes() == b.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("spp,kw", [(512, dict(sampler=abi.RT_SAMPLER_ZERO_TWO)),
                                    (288, dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=16)),
                                    (256, dict(sampler=abi.RT_SAMPLER_ZERO_TWO)),
                                    (1, dict(sampler=abi.RT_SAMPLER_ZERO_TWO)),
                                    (7, dict(sampler=abi.RT_SAMPLER_STRATIFIED, sampler_arg=7))])
def test_pixel_samplers_at_small_and_large_spp(spp, kw, scene_cache):
    """Above 256 spp the per-pixel samples are made in global memory, up to it in LDS."""
    _, cam, fr = scene_cache("random")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h = 9, 5
    a = lib.generate_rays(cam(w, h), lib.render_opts(w, h, spp, 4, seed=4, **kw), spp + 3, w * h * spp - spp - 3)
    b = orc.generate_rays(cam(w, h), orc.render_opts(w, h, spp, 4, seed=4, **kw), spp + 3, w * h * spp - spp - 3)
    assert a.tobytes() == b.tobytes()
    img, _ = g.render(cam(w, h), lib.render_opts(w, h, spp, 4, seed=4, **kw))
    ref, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 4, seed=4, **kw))
    assert img.tobytes() == ref.tobytes()
