"""Halton camera samples on the GPU against the oracle, bit for bit: generated rays,
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
