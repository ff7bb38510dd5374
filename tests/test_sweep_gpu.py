"""A sweep over seeded scenes and settings: each case draws its own scene size, leaf size, split
method, node source, number of stack levels in LDS and ray set, and compares the HIP path with
the oracle bit for bit (closest hit, any hit, traversal counters, a small frame with and without
direct lighting).  The cases are few and fixed (seeds below), so a failure names its case."""
import os

import numpy as np
import pytest

from raytracer_amd import abi, lib
from raytracer_amd.scene import look_at, perspective_camera
from tests import orc
from tests.conftest import random_rays, set_node_source
from tests.mixed import mixed_scene
from tests.test_parity_gpu import assert_hits_equal

pytestmark = pytest.mark.gpu


def awkward_rays(wb, nodes, seed):
    """Rays that sit on the planes the slab test divides by: origins on the faces of the world
    box and of node boxes, directions with one or two zero components (also negative zero),
    and rays that end exactly on such a plane."""
    rng = np.random.default_rng(seed)
    n = 4096
    rays = random_rays(n, wb[:3], wb[3:], seed, finite_fraction=0.3, axis_fraction=0.3)
    boxes = np.concatenate([wb[None, :], np.concatenate([nodes["lo"], nodes["hi"]], axis=1)[:64]])
    pick = boxes[rng.integers(0, len(boxes), n)]
    axis = rng.integers(0, 3, n)
    side = rng.integers(0, 2, n)
    on_plane = rng.random(n) < 0.5
    o = rays["o"].copy()
    o[np.arange(n), axis] = np.where(on_plane, pick[np.arange(n), axis + 3 * side], o[np.arange(n), axis])
    rays["o"] = o
    d = rays["d"].copy()
    flip = rng.random(n) < 0.2
    d[flip] = np.where(d[flip] == 0.0, np.float32(-0.0), d[flip])  # negative zeros
    rays["d"] = d
    # rays that end on a plane of the world box along their dominant axis
    k = np.argmax(np.abs(d), axis=1)
    target = np.where(d[np.arange(n), k] > 0, wb[3 + k], wb[k])
    t_end = (target - o[np.arange(n), k]) / d[np.arange(n), k]
    ends = (rng.random(n) < 0.2) & (t_end > 0)
    rays["t_max"] = np.where(ends, t_end.astype(np.float32), rays["t_max"])
    return rays


CASES = [  # seed, n_each, max_prims, split, nodes, stack levels in LDS (0: the library's choice)
    (101, 3, 1, abi.RT_SPLIT_SAH, "lds", 0),
    (102, 17, 2, abi.RT_SPLIT_MIDDLE, "global", 0),
    (103, 90, 7, abi.RT_SPLIT_EQUAL_COUNTS, "lds", 3),
    (104, 40, 255, abi.RT_SPLIT_HLBVH, "global", 2),
    (105, 150, 1, abi.RT_SPLIT_HLBVH, "lds", 0),
    (106, 8, 4, abi.RT_SPLIT_SAH, "global", 1),
    (107, 60, 3, abi.RT_SPLIT_MIDDLE, "lds", 5),
    (108, 250, 4, abi.RT_SPLIT_SAH, "global", 0),
]


def _drawn(seed):
    r = np.random.default_rng(seed)
    return (seed, int(r.integers(2, 200)), int(r.choice([1, 2, 3, 4, 5, 8, 16, 64, 255])), int(r.integers(0, 4)),
            str(r.choice(["lds", "global"])), int(r.choice([0, 0, 1, 2, 3, 6])))


CASES += [_drawn(seed) for seed in range(109, 133)]
# the same drawn cases that came out "global", once more with four-box records in every launch
CASES += [c[:4] + ("quads",) + c[5:] for c in CASES if c[4] == "global"]
# a longer soak on request: RT355_SWEEP_EXTRA=<n> draws n more cases
CASES += [_drawn(seed) for seed in range(1000, 1000 + int(os.environ.get("RT355_SWEEP_EXTRA", "0")))]


@pytest.mark.parametrize("seed,n_each,max_prims,split,nodes,stack_lds", CASES)
def test_seeded_case(monkeypatch, seed, n_each, max_prims, split, nodes, stack_lds):
    import torch
    set_node_source(monkeypatch, nodes)
    if stack_lds:
        monkeypatch.setenv("RT355_STACK_LDS", str(stack_lds))
    fr = mixed_scene(seed=seed, n_each=n_each, with_normals=bool(seed & 1), with_media=bool(seed & 2)).freeze()
    o = orc.OracleScene(fr, max_prims, split)
    g = lib.Scene(fr, max_prims, split)
    assert g.nodes().tobytes() == o.nodes().tobytes()
    assert (g.prim_order() == o.prim_order()).all()
    wb = o.world_bound()
    lo, hi = np.maximum(wb[:3], -40), np.minimum(wb[3:], 40)
    for rays, what in ((random_rays(30000 + seed, lo, hi, seed), "random"),
                       (awkward_rays(np.concatenate([lo, hi]), o.nodes(), seed), "awkward")):
        ref, ref_counters = o.intersect(rays, counters=True)
        assert_hits_equal(g.intersect(rays), ref, "case %d %s" % (seed, what))
        assert (g.does_intersect(rays) == o.occluded(rays)).all()
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
        d_hits = torch.empty((len(rays), 8), dtype=torch.float32, device="cuda")
        counters = g.intersect_count(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
        assert_hits_equal(d_hits.cpu().numpy().view(abi.hit_dtype).reshape(-1), ref,
                          "case %d %s (counting)" % (seed, what))
        assert (counters == ref_counters).all(), (counters, ref_counters)
    w, h = 72 + seed % 5, 40 + seed % 3  # not multiples of the 32-pixel tiles
    cam = perspective_camera(w, h, 50.0, look_at((0.0, 3.0, 24.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)),
                             lens_radius=0.05 * (seed % 2), focal_distance=24.0, shutter=(0.0, 1.0))
    for flags in (0, abi.RT_RENDER_DIRECT_LIGHT):
        img, st = g.render(cam, lib.render_opts(w, h, 3, 12, seed=seed, flags=flags))
        ref_img, rst = o.render(cam, orc.render_opts(w, h, 3, 12, seed=seed, flags=flags))
        assert img.tobytes() == ref_img.tobytes(), "case %d frame, flags %d" % (seed, flags)
        assert st.segments == rst.segments
