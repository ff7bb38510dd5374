"""The HIP film (gathered FilmTile::add_sample + write_image normalisation) against the
oracle's, bit for bit: every filter, frames rendered in several batches, raw shards."""
import numpy as np
import pytest
import torch

from raytracer_amd import abi, lib
from tests import orc
from tests.test_film_cpu import FILTERS


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(FILTERS))
def test_film_bit_exact(name, scene_cache):
    _, cam, fr = scene_cache("week")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp, f = 72, 40, 3, FILTERS[name]
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 10, seed=6, filter=f))
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 10, seed=6, filter=f))
    assert img.tobytes() == ref.tobytes()
    assert st.segments == rst.segments


@pytest.mark.gpu
def test_film_batches_and_shards(scene_cache):
    _, cam, fr = scene_cache("cornell")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp, f = 96, 64, 5, FILTERS["mitchell"]
    ref, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=3, filter=f))
    # one sample per batch: 6 tiles x 1024 pixels x 1 sample
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 8, seed=3, filter=f, max_paths=6 * 1024))
    assert st.batches == spp and img.tobytes() == ref.tobytes()
    # raw shards summed on the device, then resolved: same film up to the order of three additions
    total = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    part = torch.zeros_like(total)
    for k in range(3):
        opts = lib.render_opts(w, h, spp, 8, seed=3, filter=f, shard=(k, 3), flags=abi.RT_RENDER_RAW_FILM)
        g.render_dev(cam(w, h), opts, part.data_ptr())
        oref, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=3, filter=f, shard=(k, 3),
                                                      flags=abi.RT_RENDER_RAW_FILM))
        torch.cuda.synchronize()
        assert part.cpu().numpy().tobytes() == oref.tobytes()
        total += part
    lib.film_resolve_dev(total.data_ptr(), w, h)
    torch.cuda.synchronize()
    assert np.allclose(total.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
