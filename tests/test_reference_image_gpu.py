"""A coarse pin against the only rendered output the reference keeps: test/rendered_week.png,
the week.toml scene at 800 x 800 and 10 000 samples per pixel from a renderer that is no longer
in its tree (fixture: tests/golden/rendered_week_blocks.npy, 50 x 50 means of 16 x 16 pixel
blocks, written by tests/golden/make_reference_assets.py).

Qualitative by nature: that renderer's random boxes, sphere positions, noise tables, seed and
tone mapping are unknown, so individual blocks differ; what must agree is the layout and the
overall light distribution.  The frame is rendered at 800 x 800, 256 spp, depth 50, clamped to
[0, 1], gamma 2 (the legacy scenes' `gamma` default), averaged over the same blocks, and
compared by correlation and by RMSE after matching the mean level."""
import os

import numpy as np
import pytest

from raytracer_amd import lib

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_week_frame_resembles_the_reference_render(scene_cache):
    _, cam, fr = scene_cache("week")
    g = lib.Scene(fr)
    img, st = g.render(cam(800, 800), lib.render_opts(800, 800, 256, 50, seed=0))
    ours = np.sqrt(np.clip(img[..., :3].astype(np.float64), 0.0, 1.0)).reshape(50, 16, 50, 16, 3).mean(axis=(1, 3))
    ref = np.load(os.path.join(GOLDEN, "rendered_week_blocks.npy")).astype(np.float64)
    assert ours.shape == ref.shape == (50, 50, 3)
    corr = np.corrcoef(ours.ravel(), ref.ravel())[0, 1]
    level = ref.mean() / ours.mean()
    rmse = np.sqrt(((ours * level - ref) ** 2).mean())
    print("reference image: correlation %.3f, mean level ratio %.3f, rmse after level match %.3f" % (corr, level, rmse))
    assert corr > 0.88, corr          # same layout: lamp, spheres, cube of spheres, green boxes
    assert 0.6 < level < 1.6, level   # same overall exposure to within the unknown tone mapping
    assert rmse < 0.15, rmse
    # the brightest block is the lamp in both, in the same place (top, left of centre)
    by, bx = np.unravel_index(ours.sum(-1).argmax(), (50, 50))
    ry, rx = np.unravel_index(ref.sum(-1).argmax(), (50, 50))
    assert abs(int(by) - int(ry)) <= 3 and abs(int(bx) - int(rx)) <= 6, ((by, bx), (ry, rx))
