"""GPU parity: every result of the HIP path (through the C ABI) is compared with
the CPU oracle on identical inputs.  Hit/miss, primitive index, t, (u,v), the
normal and the face flags must be bit-identical; radiance of individual paths
and whole images must be bit-identical too, because both sides evaluate the
same fp32 expressions in the same order."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, lib
from tests import orc
from tests.conftest import scene_with_hooks, random_rays, set_node_source
from tests.mixed import mixed_scene

pytestmark = pytest.mark.gpu


def assert_hits_equal(a, b, what=""):
    for f in ("prim", "flags"):
        bad = np.nonzero(a[f] != b[f])[0]
        assert len(bad) == 0, "%s: %s differs at %d rays, first %s" % (what, f, len(bad), bad[:5])
    for f in ("t", "u", "v", "n"):
        x, y = a[f].view(np.uint32), b[f].view(np.uint32)
        bad = np.nonzero((x != y).reshape(len(a), -1).any(axis=1))[0]
        assert len(bad) == 0, "%s: %s differs at %d rays, first %s: %s vs %s" % (
            what, f, len(bad), bad[:5], a[f][bad[:3]], b[f][bad[:3]])


def scene_rays(o, cam, n_cam=1 << 15, n_rand=1 << 15, seed=3):
    wb = o.world_bound()
    lo, hi = np.maximum(wb[:3], -2000), np.minimum(wb[3:], 2000)
    opts = orc.render_opts(256, 128, 1)
    prim = orc.generate_rays(cam(256, 128), opts, 0, min(n_cam, 256 * 128))
    return np.concatenate([prim, random_rays(n_rand, lo, hi, seed)])


@pytest.mark.parametrize("name", ["random", "cornell", "week", "pbrt"])
@pytest.mark.parametrize("split", [abi.RT_SPLIT_SAH, abi.RT_SPLIT_HLBVH])
def test_intersect_named_scenes(scene_cache, name, split):
    sd, cam, fr = scene_cache(name)
    o = orc.OracleScene(fr, 4, split)
    g = lib.Scene(fr, 4, split)
    assert g.nodes().tobytes() == o.nodes().tobytes()
    rays = scene_rays(o, cam)
    ref = o.intersect(rays)
    got = g.intersect(rays)
    assert (ref["prim"] != abi.RT_MISS).mean() > 0.2
    assert_hits_equal(got, ref, name)
    occ = g.does_intersect(rays)
    assert (occ == o.occluded(rays)).all()
    assert ((ref["prim"] != abi.RT_MISS) == (occ != 0)).all()


@pytest.mark.parametrize("split", range(4))
@pytest.mark.parametrize("max_prims", [1, 4, 32])
def test_intersect_mixed_scene_all_builders(split, max_prims):
    fr = mixed_scene(seed=5 + split, n_each=60).freeze()
    o = orc.OracleScene(fr, max_prims, split)
    g = lib.Scene(fr, max_prims, split)
    assert g.nodes().tobytes() == o.nodes().tobytes()
    assert (g.prim_order() == o.prim_order()).all()
    wb = o.world_bound()
    rays = random_rays(1 << 16, np.maximum(wb[:3], -30), np.minimum(wb[3:], 30), 11 + split)
    ref = o.intersect(rays)
    kinds = set(np.unique(ref["flags"][ref["prim"] != abi.RT_MISS] & 0xFF))
    assert kinds == set(range(7)), kinds  # every primitive kind is hit by some ray
    assert_hits_equal(g.intersect(rays), ref, "mixed")
    assert (g.does_intersect(rays) == o.occluded(rays)).all()


def test_counters_match_oracle(scene_cache):
    import torch
    sd, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    g = lib.Scene(fr)
    rays = scene_rays(o, cam, 1 << 14, 1 << 14)
    ref, cnt = o.intersect(rays, counters=True)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((len(rays), 8), dtype=torch.float32, device="cuda")
    got_cnt = g.intersect_count(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
    assert (got_cnt == cnt).all(), (got_cnt, cnt)
    got = d_hits.cpu().numpy().view(abi.hit_dtype).reshape(-1)
    assert_hits_equal(got, ref, "count variant")
    # same rays through the asynchronous device-pointer entry point on torch's stream
    d_hits.zero_()
    g.intersect_dev(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_hits_equal(d_hits.cpu().numpy().view(abi.hit_dtype).reshape(-1), ref, "dev variant")


def test_edge_cases():
    fr = mixed_scene(seed=2, n_each=10).freeze()
    g = lib.Scene(fr)
    o = orc.OracleScene(fr)
    # empty batch
    assert len(g.intersect(np.zeros(0, dtype=abi.ray_dtype))) == 0
    # ragged batch sizes around the wave and block sizes
    wb = o.world_bound()
    for n in (1, 63, 64, 65, 255, 257, 1000):
        rays = random_rays(n, wb[:3], wb[3:], n)
        assert_hits_equal(g.intersect(rays), o.intersect(rays), "n=%d" % n)
    # degenerate rays: zero t_max, negative t_max, NaN direction, zero direction
    rays = random_rays(8, wb[:3], wb[3:], 1)
    rays["t_max"][:2] = 0.0
    rays["t_max"][2] = -1.0
    rays["d"][3] = np.nan
    rays["d"][4] = 0.0
    assert_hits_equal(g.intersect(rays), o.intersect(rays), "degenerate")
    # rays along the shared edge of the watertight triangle pair: exactly one hit
    rays = np.zeros(64, dtype=abi.ray_dtype)
    s = np.linspace(-2.9, 2.9, 64, dtype=np.float32)
    rays["o"] = np.stack([s, s, np.full(64, 5.0, np.float32)], axis=1)
    rays["d"] = (0.0, 0.0, -1.0)
    rays["t_max"] = np.inf
    got, ref = g.intersect(rays), o.intersect(rays)
    assert_hits_equal(got, ref, "shared edge")


def test_empty_scene():
    from raytracer_amd.scene import SceneData
    fr = SceneData().freeze()
    g = lib.Scene(fr)
    assert g.info().n_nodes == 0
    wb = g.world_bound()
    assert (wb[:3] > wb[3:]).all()  # Bounds3::ZERO
    rays = random_rays(100, (-1, -1, -1), (1, 1, 1), 0)
    hits = g.intersect(rays)
    assert (hits["prim"] == abi.RT_MISS).all() and np.isinf(hits["t"]).all()
    assert not g.does_intersect(rays).any()


def test_generate_rays_match(scene_cache):
    for name in ("random", "week", "pbrt"):
        sd, cam, fr = scene_cache(name)
        c = cam(320, 200)
        opts = orc.render_opts(320, 200, 4, seed=9)
        ref = orc.generate_rays(c, opts, 1000, 50000)
        got = lib.generate_rays(c, opts, 1000, 50000)
        assert ref.tobytes() == got.tobytes(), name


@pytest.mark.parametrize("name,w,h,spp", [("random", 160, 96, 4), ("cornell", 96, 96, 4),
                                          ("week", 128, 128, 4), ("pbrt", 96, 96, 4)])
def test_render_bit_exact(scene_cache, name, w, h, spp):
    sd, cam, fr = scene_cache(name)
    o, g = orc.OracleScene(fr), lib.Scene(fr)
    c = cam(w, h)
    ref, rs = o.render(c, orc.render_opts(w, h, spp, 50, seed=1, flags=abi.RT_RENDER_COUNTERS))
    got, gs = g.render(c, lib.render_opts(w, h, spp, 50, seed=1, flags=abi.RT_RENDER_COUNTERS))
    assert gs.paths == rs.paths == w * h * spp
    assert gs.segments == rs.segments
    assert gs.nodes_visited == rs.nodes_visited
    assert list(gs.prims_tested) == list(rs.prims_tested)
    assert np.isfinite(got).all()
    assert got.tobytes() == ref.tobytes(), "max abs diff %g" % np.abs(got - ref).max()
    # small batches (several sample ranges per pixel) give the same image
    got2, _ = g.render(c, lib.render_opts(w, h, spp, 50, seed=1, max_paths=w * h))
    assert got2.tobytes() == ref.tobytes()


def test_render_mixed_and_sharded():
    fr = mixed_scene(seed=4, n_each=30).freeze()
    from raytracer_amd.scene import look_at, perspective_camera
    w, h, spp = 100, 70, 3
    c = perspective_camera(w, h, 50.0, look_at((0, 3, -30), (0, 0, 0), (0, 1, 0)), lens_radius=0.2,
                           focal_distance=30.0, shutter=(0.0, 1.0))
    o, g = orc.OracleScene(fr), lib.Scene(fr)
    ref, _ = o.render(c, orc.render_opts(w, h, spp, 12, seed=7))
    got, _ = g.render(c, lib.render_opts(w, h, spp, 12, seed=7))
    assert got.tobytes() == ref.tobytes()
    # tile sharding: the shards' images add up to the whole frame, bit for bit
    total = np.zeros_like(got)
    for k in range(3):
        part, st = g.render(c, lib.render_opts(w, h, spp, 12, seed=7, shard=(k, 3)))
        assert ((total[..., 3] == 0) | (part[..., 3] == 0)).all()  # disjoint ownership
        total += part
    assert total.tobytes() == ref.tobytes()


def test_full_size_properties(scene_cache):
    """Full-size frame (week scene, 1920x1080): checked through properties that do
    not need the oracle at full size -- a sample of its paths against the oracle,
    determinism, and independence of the batch size."""
    sd, cam, fr = scene_cache("week")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp = 1920, 1080, 2
    c = cam(w, h)
    a, sa = g.render(c, lib.render_opts(w, h, spp, 50, seed=0))
    b, sb = g.render(c, lib.render_opts(w, h, spp, 50, seed=0, max_paths=1 << 21))
    assert a.tobytes() == b.tobytes() and sa.segments == sb.segments
    assert np.isfinite(a).all() and (a[..., 3] == 1).all()
    # 200 pixels spread over the frame: oracle radiance of their paths, summed in order
    rng = np.random.default_rng(0)
    opts = orc.render_opts(w, h, spp, 50, seed=0)
    for pix in rng.integers(0, w * h, 200):
        rgb, _ = o.path_radiance(c, opts, int(pix) * spp, spp)
        acc = np.zeros(3, np.float32)
        for s in range(spp):
            acc = acc + rgb[s]
        exp = acc / np.float32(spp)
        assert (a[pix // w, pix % w, :3] == exp).all(), pix


@pytest.mark.gpu
@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_streaming_paths(monkeypatch, nodes):
    """Rays enter and leave a wave one by one: every batch size around the 64-entry blocks,
    rays that miss the root box (lanes that never traverse), and both node sources."""
    set_node_source(monkeypatch, nodes)
    fr = mixed_scene(seed=9, n_each=60).freeze()
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    wb = o.world_bound()
    for n in (2, 31, 32, 33, 127, 128, 129, 4095, 4097, 70001):
        rays = random_rays(n, wb[:3], wb[3:], 100 + n)
        # every third ray starts far away and points outwards: misses the root
        rays["o"][::3] = wb[3:] + 50.0
        rays["d"][::3] = np.abs(rays["d"][::3]) + 0.1
        assert_hits_equal(g.intersect(rays), o.intersect(rays), "%s n=%d" % (nodes, n))
        assert (g.does_intersect(rays) == o.occluded(rays)).all()
    # a tree that is a single leaf (no interior record) and a single primitive
    from raytracer_amd.scene import SceneData
    s = SceneData()
    s.lambertian((1, 1, 1))
    s.add_sphere((0, 0, -3), 1.0)
    fr1 = s.freeze()
    g1, o1 = lib.Scene(fr1), orc.OracleScene(fr1)
    rays = random_rays(1000, np.float32([-2, -2, -5]), np.float32([2, 2, 1]), 5)
    assert_hits_equal(g1.intersect(rays), o1.intersect(rays), "single leaf")
    img, st = g1.render(_cam(64, 48), lib.render_opts(64, 48, 3, 8, seed=3))
    ref, rst = o1.render(_cam(64, 48), orc.render_opts(64, 48, 3, 8, seed=3))
    assert img.tobytes() == ref.tobytes() and st.segments == rst.segments


@pytest.mark.gpu
@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
@pytest.mark.parametrize("direct", [False, True])
def test_bounce_as_one_kernel_or_two(monkeypatch, scene_cache, nodes, direct):
    """A bounce is one fused launch (trees in LDS) or a trace launch and a shade launch (trees in
    global memory); RT355_SPLIT forces either form on either node source.  Frames, segment counts
    per bounce and traversal counters must not depend on the form, and equal the oracle's."""
    from raytracer_amd import abi
    set_node_source(monkeypatch, nodes)
    name, cam, fr = scene_cache("cornell" if direct else "week")
    flags = abi.RT_RENDER_DIRECT_LIGHT if direct else 0
    w, h, spp, depth = 96, 64, 5, 12
    o = orc.OracleScene(fr)
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, depth, seed=4, flags=flags | abi.RT_RENDER_COUNTERS))
    per_bounce = None
    for split in (0, 1):
        # RT355_LOOP_BELOW=0: every bounce is a per-bounce launch, so the form under test runs throughout
        g = scene_with_hooks(monkeypatch, fr, RT355_SPLIT=split, RT355_LOOP_BELOW=0)
        img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=4, flags=flags))
        assert img.tobytes() == ref.tobytes(), (nodes, split)
        assert st.segments == rst.segments, (nodes, split)
        assert per_bounce is None or list(st.bounce_rays) == per_bounce
        per_bounce = list(st.bounce_rays)
        img2, cst = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=4, flags=flags | abi.RT_RENDER_COUNTERS))
        assert img2.tobytes() == ref.tobytes()
        assert cst.nodes_visited == rst.nodes_visited and list(cst.prims_tested) == list(rst.prims_tested), (nodes, split)
        g.close()


def _cam(w, h):
    from raytracer_amd.scene import look_at, perspective_camera
    return perspective_camera(w, h, 40.0, look_at((0.0, 0.0, 2.0), (0.0, 0.0, -3.0), (0.0, 1.0, 0.0)))


@pytest.mark.gpu
@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_stack_levels_beyond_lds(monkeypatch, nodes, scene_cache):
    """With only two stack levels in LDS most pushes land in the per-lane private array."""
    monkeypatch.setenv("RT355_STACK_LDS", "2")
    set_node_source(monkeypatch, nodes)
    name, cam, fr = scene_cache("week")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    wb = o.world_bound()
    rays = random_rays(50000, wb[:3], wb[3:], 77)
    got, ref = g.intersect(rays), o.intersect(rays)
    assert_hits_equal(got, ref, "spill " + nodes)
    assert (g.does_intersect(rays) == o.occluded(rays)).all()
    img, st = g.render(cam(96, 64), lib.render_opts(96, 64, 4, 20, seed=2))
    ref_img, rst = o.render(cam(96, 64), orc.render_opts(96, 64, 4, 20, seed=2))
    assert img.tobytes() == ref_img.tobytes() and st.segments == rst.segments


@pytest.mark.gpu
@pytest.mark.parametrize("stack_lds", [0, 2])
def test_tree_in_lds_with_the_most_primitives_it_takes(monkeypatch, stack_lds):
    """32 767 primitives is the most a tree traversed from LDS may index (15-bit leaf references,
    16-bit stack entries read back sign-extended).  129 groups of up to 255 concentric spheres make
    a few hundred large leaves (centroids that differ by rounding only), so the records fit in LDS
    while the leaf references run up to the top of the 15 bits."""
    from raytracer_amd.scene import SceneData, look_at, perspective_camera
    if stack_lds:
        monkeypatch.setenv("RT355_STACK_LDS", str(stack_lds))
    rng = np.random.default_rng(5)
    n = 0x7fff
    centres = np.repeat(rng.uniform(-20, 20, (129, 3)), 255, axis=0)[:n]
    sd = SceneData(background=(0.6, 0.7, 0.9))
    grey = sd.lambertian((0.6, 0.6, 0.6))
    sd.add_sphere_block(centres, rng.uniform(0.05, 2.0, n), grey)
    fr = sd.freeze()
    g, o = lib.Scene(fr, 4, abi.RT_SPLIT_SAH), orc.OracleScene(fr, 4, abi.RT_SPLIT_SAH)
    assert g.info().n_prims == n and g.info().n_nodes < 2000  # 64 KB of two-child records at most
    assert g.nodes().tobytes() == o.nodes().tobytes()
    wb = o.world_bound()
    rays = random_rays(40000, wb[:3], wb[3:], 11)
    ref, ref_counters = o.intersect(rays, counters=True)
    assert_hits_equal(g.intersect(rays), ref, "32767 primitives")
    hit = ref["prim"] != 0xffffffff
    assert o.prim_order()[-200:].min() < 0x7fff and np.isin(ref["prim"][hit], o.prim_order()[-2000:]).any()  # the last leaves are reached
    assert (g.does_intersect(rays) == o.occluded(rays)).all()
    import torch
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((len(rays), 8), dtype=torch.float32, device="cuda")
    counters = g.intersect_count(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
    assert (counters == ref_counters).all()
    cam = perspective_camera(64, 48, 60.0, look_at((0.0, 5.0, 45.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))
    img, st = g.render(cam, lib.render_opts(64, 48, 2, 8, seed=3))
    ref_img, rst = o.render(cam, orc.render_opts(64, 48, 2, 8, seed=3))
    assert img.tobytes() == ref_img.tobytes() and st.segments == rst.segments


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name", ["week", "cornell", "random"])
@pytest.mark.parametrize("loop_from", [0, 1, 4])
def test_paths_followed_in_one_launch(monkeypatch, scene_cache, scene_name, loop_from):
    """From bounce `loop_from` on, one launch follows the paths to their end (a wave feeds the
    survivors of its shading batches back to its own lanes): same image, same segments per
    bounce and same traversal counters as one launch per bounce, and as the oracle."""
    _, cam, fr = scene_cache(scene_name)
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp, depth = 160, 96, 3, 50
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, depth, seed=5, flags=abi.RT_RENDER_COUNTERS))
    per_bounce = scene_with_hooks(monkeypatch, fr, RT355_LOOP_BELOW=0)  # one launch per bounce
    base, bst = per_bounce.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=5, flags=abi.RT_RENDER_COUNTERS))
    auto, ast = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=5))  # the library decides (small frame)
    assert ast.trace_launches < depth and auto.tobytes() == base.tobytes() and ast.segments == bst.segments
    g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_FROM=loop_from)
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=5, flags=abi.RT_RENDER_COUNTERS))
    assert st.trace_launches == loop_from + 1 and bst.trace_launches == depth
    assert img.tobytes() == ref.tobytes() == base.tobytes()
    assert st.segments == rst.segments and list(st.bounce_rays) == list(bst.bounce_rays)
    assert st.nodes_visited == rst.nodes_visited and list(st.prims_tested) == list(rst.prims_tested)
    # depth limits: the last bounce allowed, and a limit of one
    for d in (1, 2, loop_from + 2):
        a, sa = g.render(cam(w, h), lib.render_opts(w, h, spp, d, seed=5))
        b, sb = o.render(cam(w, h), orc.render_opts(w, h, spp, d, seed=5))
        assert a.tobytes() == b.tobytes() and sa.segments == sb.segments


@pytest.mark.gpu
def test_one_launch_kernel_with_film_sampler_and_shards(monkeypatch, scene_cache):
    """The kernel that follows paths to their end under everything else a render can ask for:
    a reconstruction filter, a low-discrepancy sampler, several sample batches and tile shards,
    with trees read from LDS and from global memory."""
    from tests.test_film_cpu import FILTERS
    _, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    w, h, spp = 96, 64, 4
    kw = dict(seed=2, filter=FILTERS["mitchell"], sampler=abi.RT_SAMPLER_SOBOL)
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 30, **kw))
    for nodes in ("lds", "global"):
        set_node_source(monkeypatch, nodes)
        for loop_from in (1, 3):
            g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_FROM=loop_from)
            img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 30, max_paths=6 * 1024 * 2, **kw))
            assert st.batches == 2 and img.tobytes() == ref.tobytes() and st.segments == rst.segments
        parts = [g.render(cam(w, h), lib.render_opts(w, h, spp, 30, seed=2, shard=(k, 3)))[0] for k in range(3)]
        whole, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 30, seed=2))
        assert (parts[0][..., :3] + parts[1][..., :3] + parts[2][..., :3]).tobytes() == whole[..., :3].tobytes()
        g.close()


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_one_launch_kernel_shares_sub_queues(monkeypatch, scene_cache, nodes):
    """A frame with more blocks per sub-queue than waves, so that blocks are claimed and a wave
    whose sub-queue is read out goes on with the others': same image and per-bounce segment
    counts as the oracle and as one launch per bounce, with the sharing on and off."""
    set_node_source(monkeypatch, nodes)
    _, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    w, h, spp, depth = 320, 200, 16, 50
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, depth, seed=9))
    g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_BELOW=0)
    base, bst = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=9))
    assert base.tobytes() == ref.tobytes() and bst.segments == rst.segments
    for loop_from in (0, 1, 3):
        for steal in ("1", "0"):
            g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_FROM=loop_from, RT355_LOOP_STEAL=steal)
            img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=9))
            assert st.trace_launches == loop_from + 1
            assert img.tobytes() == ref.tobytes(), (loop_from, steal)
            assert st.segments == rst.segments and list(st.bounce_rays) == list(bst.bounce_rays)


def test_switch_bounce_is_remembered_between_frames(scene_cache):
    """The first frame of a shape asks the device for its queue sizes to decide where the
    one-launch kernel takes over; later frames of that shape switch at the same bounce without
    the round trips: same launches, same image, also after a frame of another shape."""
    _, cam, fr = scene_cache("week")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    w, h, spp = 160, 96, 3
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 50, seed=8))
    first, fst = g.render(cam(w, h), lib.render_opts(w, h, spp, 50, seed=8))
    other, _ = g.render(cam(96, 64), lib.render_opts(96, 64, spp, 50, seed=8))
    again, ast = g.render(cam(w, h), lib.render_opts(w, h, spp, 50, seed=8))
    third, tst = g.render(cam(w, h), lib.render_opts(w, h, spp, 50, seed=9))
    assert first.tobytes() == again.tobytes() == ref.tobytes() and fst.segments == ast.segments == rst.segments
    assert 1 < fst.trace_launches < 50
    assert tst.trace_launches == fst.trace_launches  # the 160 x 96 shape's bounce, found by its first frame
    ref9, _ = o.render(cam(w, h), orc.render_opts(w, h, spp, 50, seed=9))
    assert third.tobytes() == ref9.tobytes()


def test_switch_bounce_is_looked_up_again_when_it_no_longer_fits(monkeypatch, scene_cache):
    """The remembered bounce belongs to a frame shape, not to a camera: when a frame of the same
    shape shrinks its queues faster (here: a camera that looks past the scene), the counts that come
    back with it show that the switch could have come earlier, and the next frame looks again."""
    from raytracer_amd.scene import look_at, perspective_camera
    _, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    w, h, spp = 160, 96, 3
    g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_BELOW=12000)
    opts = lib.render_opts(w, h, spp, 50, seed=8)
    first, fst = g.render(cam(w, h), opts)
    assert fst.bounce_rays[2] >= 12000 > fst.bounce_rays[4] and fst.trace_launches == 5  # bounces 0-3, then the rest
    away = perspective_camera(w, h, 40.0, look_at((478.0, 278.0, -600.0), (478.0, 278.0, -1200.0), (0.0, 1.0, 0.0)))
    a, ast = g.render(away, opts)
    assert ast.bounce_rays[2] < 12000 and ast.trace_launches == 5  # most paths have left by bounce 2; switched where the frame before did
    b, bst = g.render(away, opts)
    assert bst.trace_launches == 3  # looked again: bounces 0 and 1, then the rest
    ref, _ = o.render(away, orc.render_opts(w, h, spp, 50, seed=8))
    assert a.tobytes() == b.tobytes() == ref.tobytes()
    again, gst = g.render(cam(w, h), opts)  # and back: the queue at bounce 2 is too large again
    assert gst.trace_launches == 3 and again.tobytes() == first.tobytes()
    last, lst = g.render(cam(w, h), opts)
    assert lst.trace_launches == 5 and last.tobytes() == first.tobytes()


def test_paths_deeper_than_a_byte_counts(monkeypatch, scene_cache):
    """The one-launch kernel carries a segment's depth through its ring beside the producer's
    findings; depths above 255 must not wrap (week.toml's fog-filled ball keeps paths alive that
    long).  Beyond 16 bits the call is refused."""
    _, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    w, h, spp, depth = 64, 48, 2, 300
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, depth, seed=6))
    g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_FROM=2)
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, depth, seed=6))
    assert st.trace_launches == 3 and st.max_bounce > 256  # some paths do get that deep
    assert img.tobytes() == ref.tobytes() and st.segments == rst.segments
    with pytest.raises(lib.RtError) as e:
        g.render(cam(w, h), lib.render_opts(w, h, spp, 1 << 16, seed=6))
    assert e.value.code == abi.RT_ERR_LIMIT


def test_spin_guard_of_the_one_launch_kernel_is_reported(monkeypatch, scene_cache):
    """A wave of the kernel that follows paths to their end gives up after a fixed number of
    feed turns and drops its segments; the call must then fail instead of returning the
    incomplete frame.  Forced here with a limit of three turns."""
    _, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    w, h, spp = 96, 64, 4
    g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_FROM=1, RT355_LOOP_TURN_LIMIT=3)
    with pytest.raises(lib.RtError) as e:
        g.render(cam(w, h), lib.render_opts(w, h, spp, 50, seed=4))
    assert e.value.code == abi.RT_ERR_LIMIT and "incomplete" in str(e.value)
    # frames enqueued without statistics are not waited for: the next call on the device reports it
    import torch
    d_img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    g.render_dev(cam(w, h), lib.render_opts(w, h, spp, 50, seed=4), d_img.data_ptr(), want_stats=False)
    with pytest.raises(lib.RtError) as e:
        g.render(cam(w, h), lib.render_opts(w, h, spp, 2, seed=4))
    assert e.value.code == abi.RT_ERR_LIMIT and "frame before" in str(e.value)
    g = scene_with_hooks(monkeypatch, fr, RT355_LOOP_FROM=1)
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 50, seed=4))
    ref, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 50, seed=4))
    assert img.tobytes() == ref.tobytes() and st.segments == rst.segments


def test_concurrent_callers(scene_cache):
    """Host threads share one scene: four trace callers (two of them on the same stream) and
    a renderer at once; every result equals the single-threaded oracle result."""
    import threading
    import torch
    sd, cam, fr = scene_cache("week")
    o = orc.OracleScene(fr)
    g = lib.Scene(fr)
    g.upload()
    wb = o.world_bound()
    lo, hi = np.maximum(wb[:3], -2000), np.minimum(wb[3:], 2000)
    batches = [random_rays(20000 + 777 * k, lo, hi, 40 + k) for k in range(4)]
    want_hits = [o.intersect(r) for r in batches]
    want_occ = [o.occluded(r) for r in batches]
    ropts = dict(width=64, height=32, spp=4, max_depth=8, seed=3)
    want_img, _ = o.render(cam(64, 32), orc.render_opts(**ropts))
    shared = torch.cuda.Stream()
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), shared, shared]
    d_rays = [torch.from_numpy(r.view(np.float32).reshape(-1, 8)).cuda() for r in batches]
    d_hits = [torch.empty((len(r), 8), dtype=torch.float32, device="cuda") for r in batches]
    d_occ = [torch.empty(len(r), dtype=torch.uint8, device="cuda") for r in batches]
    torch.cuda.synchronize()
    failures = []

    def tracer(k):
        try:
            for _ in range(6):
                d_hits[k].zero_()
                d_occ[k].zero_()
                torch.cuda.synchronize()
                st = streams[k].cuda_stream
                g.intersect_dev(d_rays[k].data_ptr(), len(batches[k]), d_hits[k].data_ptr(), st)
                g.occluded_dev(d_rays[k].data_ptr(), len(batches[k]), d_occ[k].data_ptr(), st)
                streams[k].synchronize()
                assert_hits_equal(d_hits[k].cpu().numpy().view(abi.hit_dtype).reshape(-1), want_hits[k], "thread %d" % k)
                assert (d_occ[k].cpu().numpy() == want_occ[k]).all()
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            failures.append((k, repr(e)))

    def renderer():
        try:
            for _ in range(4):
                img, _ = g.render(cam(64, 32), lib.render_opts(**ropts))
                assert img.tobytes() == want_img.tobytes()
        except BaseException as e:  # noqa: BLE001
            failures.append(("render", repr(e)))

    threads = [threading.Thread(target=tracer, args=(k,)) for k in range(4)] + [threading.Thread(target=renderer)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures
