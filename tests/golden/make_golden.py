#!/usr/bin/env python3
"""Writes the golden fixtures of this directory from the CPU oracle.

The reference holds no vectors for ray traversal / intersection and cannot be
built here (Rust), so these vectors are minted from the oracle restatement and
guard it (and the HIP path) against regressions; they do not pin the oracle to
the reference.  Run from the repository root: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from raytracer_amd import abi, scenes  # noqa: E402
from tests import orc  # noqa: E402
from tests.conftest import random_rays  # noqa: E402
from tests.mixed import mixed_scene  # noqa: E402


def rays_for(o, cam, n_cam, n_rand, seed):
    wb = o.world_bound()
    lo, hi = np.maximum(wb[:3], -2000), np.minimum(wb[3:], 2000)
    prim = orc.generate_rays(cam(128, 64), orc.render_opts(128, 64, 1), 0, n_cam)
    return np.concatenate([prim, random_rays(n_rand, lo, hi, seed)])


def main():
    cases = {"mixed": (mixed_scene(seed=1, n_each=40), None)}
    for name in ("random", "cornell", "week", "pbrt"):
        cases[name] = scenes.BUILDERS[name]()
    for name, (sd, cam) in cases.items():
        o = orc.OracleScene(sd.freeze())
        if cam is None:
            wb = o.world_bound()
            rays = random_rays(6000, np.maximum(wb[:3], -30), np.minimum(wb[3:], 30), 17)
        else:
            rays = rays_for(o, cam, 3000, 3000, 23)
        hits, cnt = o.intersect(rays, counters=True)
        out = {"rays": rays.view(np.float32).reshape(-1, 8), "hits": hits.view(np.uint32).reshape(-1, 8),
               "occluded": o.occluded(rays), "counters": cnt, "n_nodes": np.uint32(o.info().n_nodes)}
        if cam is not None:
            img, st = o.render(cam(64, 48), orc.render_opts(64, 48, 2, 20, seed=11))
            out["image"] = img
            out["segments"] = np.uint64(st.segments)
            # the same frame with the lights sampled (emissive rectangles; test.pbrt's distant light)
            img, st = o.render(cam(64, 48), orc.render_opts(64, 48, 2, 20, seed=11, flags=abi.RT_RENDER_DIRECT_LIGHT))
            out["image_direct"] = img
            out["segments_direct"] = np.uint64(st.segments)
            out["occlusion_direct"] = np.uint64(st.occlusion_rays)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, len(rays), "rays,", int((hits["prim"] != 0xFFFFFFFF).sum()), "hits")


if __name__ == "__main__":
    main()
