"""Checker for full-size frames: sampled pixels of a rendered image re-traced by the CPU
oracle, path by path, and compared bit for bit.  Test infrastructure (the measuring scripts
call it after the timed region); nothing of the product imports it."""
import numpy as np

from tests import orc


def pixels_different(frozen, leaf_size, split, cam, width, height, spp, depth, image, count, seed=0, **sampler):
    """Number of ``count`` randomly chosen pixels of ``image`` [h, w, 4] whose RGB differs from
    the oracle's mean over the pixel's ``spp`` paths (added in sample order, as k_resolve does)."""
    o = orc.OracleScene(frozen, leaf_size, split)
    opts = orc.render_opts(width, height, spp, depth, seed=seed, **sampler)
    rng = np.random.default_rng(0)
    bad = 0
    for pix in rng.integers(0, width * height, count):
        rgb, _ = o.path_radiance(cam, opts, int(pix) * spp, spp)
        acc = np.zeros(3, np.float32)
        for s in range(spp):
            acc = acc + rgb[s]
        bad += int(not (image[pix // width, pix % width, :3] == acc / np.float32(spp)).all())
    return bad
