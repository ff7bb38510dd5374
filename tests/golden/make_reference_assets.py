#!/usr/bin/env python3
"""Turns the two image files the reference holds under test/ into fixtures (run in the build
container, where /root/reference exists; the GPU box only sees the outputs):

  earthmap_rgb8.npz          test/earthmap.jpg (the Image texture of test/week.toml:101 and
                             test/globe.toml) decoded once with PIL to RGB8, 512 x 1024 x 3
  rendered_week_blocks.npy   test/rendered_week.png (800 x 800, the only rendered output the
                             reference keeps: week.toml at 10 000 spp by a renderer that is no
                             longer in the tree) as 50 x 50 means of 16 x 16 pixel blocks,
                             values / 255, float32
"""
import os
import sys

import numpy as np
from PIL import Image

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/test"
HERE = os.path.dirname(os.path.abspath(__file__))

earth = np.asarray(Image.open(os.path.join(REF, "earthmap.jpg")).convert("RGB"), dtype=np.uint8)
assert earth.shape == (512, 1024, 3)
np.savez_compressed(os.path.join(HERE, "earthmap_rgb8.npz"), rgb8=earth)

week = np.asarray(Image.open(os.path.join(REF, "rendered_week.png")).convert("RGB"), dtype=np.float64) / 255.0
assert week.shape == (800, 800, 3)
np.save(os.path.join(HERE, "rendered_week_blocks.npy"),
        week.reshape(50, 16, 50, 16, 3).mean(axis=(1, 3)).astype(np.float32))
print("earthmap", earth.shape, "blocks", (50, 50, 3))
