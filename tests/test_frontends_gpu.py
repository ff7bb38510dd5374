"""The scene front-ends on the GPU: the reference's own scene files (test/week.toml, cornell.toml,
random.toml, test.pbrt -- kept as fixtures under tests/golden/scenes/) loaded by
raytracer_amd.frontends and handed to the library through the C ABI.  Closest hits, any-hit
bytes and small frames (with and without the lights sampled) are compared with the committed
golden vectors of tests/golden/*.npz -- no oracle runs here."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, frontends, lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = [("week.toml", "week"), ("cornell.toml", "cornell"), ("random.toml", "random"), ("test.pbrt", "pbrt")]


@pytest.mark.parametrize("file,golden", FILES)
def test_loaded_files_reproduce_the_golden_vectors(file, golden):
    sd, cam, settings = frontends.load(os.path.join(HERE, "golden", "scenes", file))
    z = np.load(os.path.join(HERE, "golden", golden + ".npz"))
    g = lib.Scene(sd.freeze())
    assert g.info().n_nodes == z["n_nodes"]
    rays = np.ascontiguousarray(z["rays"]).view(abi.ray_dtype).reshape(-1)
    hits = np.ascontiguousarray(z["hits"]).view(abi.hit_dtype).reshape(-1)
    assert g.intersect(rays).tobytes() == hits.tobytes()
    assert (g.does_intersect(rays) == z["occluded"]).all()
    for flags, key in ((0, ""), (abi.RT_RENDER_DIRECT_LIGHT, "_direct")):
        img, st = g.render(cam(64, 48), lib.render_opts(64, 48, 2, 20, seed=11, flags=flags))
        want = z["image" + key]
        if file.endswith(".pbrt"):
            # the file's camera is the inverse of its accumulated LookAt matrix and differs from
            # the hand-built one in the last bits: the frames agree pixel for pixel except where a
            # sample crosses an edge
            close = np.isclose(img, want, rtol=1e-3, atol=1e-4).all(axis=-1)
            assert close.mean() > 0.97, close.mean()
            assert abs(int(st.segments) - int(z["segments" + key])) < 0.01 * int(z["segments" + key])
        else:
            assert img.tobytes() == want.tobytes()
            assert st.segments == z["segments" + key]
            if key:
                assert st.occlusion_rays == z["occlusion_direct"]


def test_the_pbrt_file_renders_with_its_own_settings():
    """test.pbrt as the file describes it: Halton sampler, both of its lights (the uniform sky and
    the distant light through RT_RENDER_DIRECT_LIGHT).  The sun-lit floor is brighter than the
    sky-lit one and warmer (more red than blue is added), and the glass ball casts a shadow."""
    sd, cam, st = frontends.load(os.path.join(HERE, "golden", "scenes", "test.pbrt"))
    assert st["skipped"] == [] and len(sd.distant) == 1
    g = lib.Scene(sd.freeze())
    w = h = 96
    kw = frontends.sampler_options(st)
    kw["spp"] = 16
    sky, s0 = g.render(cam(w, h), lib.render_opts(w, h, max_depth=st["max_depth"], seed=1, **kw))
    sun, s1 = g.render(cam(w, h), lib.render_opts(w, h, max_depth=st["max_depth"], seed=1,
                                                  flags=abi.RT_RENDER_DIRECT_LIGHT, **kw))
    assert s0.occlusion_rays == 0 and s1.occlusion_rays > w * h
    gain = (sun - sky)[..., :3].reshape(-1, 3).mean(0)
    assert gain.min() > 0 and gain[0] > 1.5 * gain[2]  # light was added, that of a 3000 K black body
    # paths of one segment: the camera rays are the same with and without the flag, so the difference
    # is the sun's light alone -- nowhere negative, most of the floor lit, and a shadow behind the ball
    sky1, _ = g.render(cam(w, h), lib.render_opts(w, h, max_depth=1, seed=1, **kw))
    sun1, _ = g.render(cam(w, h), lib.render_opts(w, h, max_depth=1, seed=1, flags=abi.RT_RENDER_DIRECT_LIGHT, **kw))
    direct = (sun1 - sky1)[..., :3]
    assert (direct >= 0).all()
    floor = sky1[..., :3].sum(-1) == 0  # a path that ends on the floor or the ball has gathered nothing
    assert floor.mean() > 0.5
    lit = direct.sum(-1) > 0.01
    assert (lit & floor).sum() > 0.7 * floor.sum()
    assert (~lit & floor).sum() > 30  # the glass ball itself (not Lambertian) and its shadow
