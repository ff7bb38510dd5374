"""The synthetic stress scene (displaced triangle grid under a cloud of small spheres: the fifth
benchmark configuration, 1 M triangles + 10 M spheres at full size) on the GPU.

Reduced size (8 k triangles + 200 k spheres: the tree no longer fits in LDS, leaf references
pass 2^15, the stack is deep): closest hit, any hit, traversal counters and a small render are
compared with the oracle bit for bit on camera rays and on incoherent bounce rays, with the
default stack split and with only four stack levels in LDS (most pushes then land in the
per-lane private array).

Full size (11 M primitives, 22 M nodes, depth 30, 1.26 GB of tables; leaf references above
2^23): properties that need no full-size oracle run -- the tree equals the oracle's node for
node, a device-built HLBVH equals the host-built one, frames do not depend on the batch size,
and the paths of 64 sampled pixels equal the oracle's radiance bit for bit."""
import numpy as np
import pytest

from raytracer_amd import abi, lib, scenes
from tests import orc

pytestmark = pytest.mark.gpu


def bounce_rays(rays, hits, seed):
    """Incoherent secondary rays: from the hit points, direction = facing normal + uniform unit vector."""
    hit = hits["prim"] != abi.RT_MISS
    o, d = rays["o"][hit].astype(np.float64), rays["d"][hit].astype(np.float64)
    n, t = hits["n"][hit].astype(np.float64), hits["t"][hit].astype(np.float64)
    rng = np.random.default_rng(seed)
    v = rng.normal(size=o.shape)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    nf = np.where(((n * d).sum(1) < 0)[:, None], n, -n)
    p = o + t[:, None] * d
    out = np.zeros(int(hit.sum()), dtype=abi.ray_dtype)
    out["o"] = (p + nf * 1e-3 * (1.0 + np.abs(p).max(axis=1, keepdims=True))).astype(np.float32)
    nd = nf + v
    out["d"] = (nd / np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-20)).astype(np.float32)
    out["t_max"] = np.inf
    out["t_max"][::7] = 50.0  # some finite segments
    return out


def assert_same_hits(got, ref, what):
    assert got.tobytes() == ref.tobytes(), "%s: %d of %d hit records differ" % (
        what, int((got.view(np.uint8).reshape(len(got), -1) != ref.view(np.uint8).reshape(len(ref), -1)).any(1).sum()),
        len(got))


@pytest.mark.parametrize("quads", [1, 2])
@pytest.mark.parametrize("stack_lds", [None, 4])
@pytest.mark.parametrize("split", [abi.RT_SPLIT_SAH, abi.RT_SPLIT_HLBVH])
def test_reduced_synthetic_scene_bit_exact(monkeypatch, scene_cache, split, stack_lds, quads):
    import torch
    if stack_lds:
        monkeypatch.setenv("RT355_STACK_LDS", str(stack_lds))
    # 1: four-box records from the third bounce on (the library's choice), 2: in every launch
    monkeypatch.setenv("RT355_QUADS", str(quads))
    _, cam, fr = scene_cache("synthetic", grid=64, n_spheres=200_000)
    g, o = lib.Scene(fr, 4, split), orc.OracleScene(fr, 4, split)
    info = g.info()
    assert info.n_prims == 2 * 64 * 64 + 200_000 and info.n_prims > 0x8000 and info.max_depth > 14
    assert g.nodes().tobytes() == o.nodes().tobytes() and (g.prim_order() == o.prim_order()).all()
    w, h = 512, 256
    primary = lib.generate_rays(cam(w, h), lib.render_opts(w, h, 1, seed=2), 0, w * h)
    ref, cnt = o.intersect(primary, counters=True)
    assert (ref["prim"] != abi.RT_MISS).mean() > 0.5
    got = g.intersect(primary)
    assert_same_hits(got, ref, "primary")
    second = bounce_rays(primary, ref, 3)
    assert len(primary) + len(second) >= 100_000
    ref2, cnt2 = o.intersect(second, counters=True)
    assert_same_hits(g.intersect(second), ref2, "bounce 1")
    kinds = set(np.unique(ref2["flags"][ref2["prim"] != abi.RT_MISS] & 0xFF))
    assert kinds == {abi.RT_SPHERE, abi.RT_TRIANGLE}
    for rays in (primary, second):
        assert (g.does_intersect(rays) == o.occluded(rays)).all()
    # traversal counters of the device-pointer entry point
    d_rays = torch.from_numpy(second.view(np.float32).reshape(-1, 8)).cuda()
    d_hits = torch.empty((len(second), 8), dtype=torch.float32, device="cuda")
    assert (g.intersect_count(d_rays.data_ptr(), len(second), d_hits.data_ptr()) == cnt2).all()
    # a small frame: image, segments and counters
    w, h, spp = 128, 64, 2
    img, st = g.render(cam(w, h), lib.render_opts(w, h, spp, 8, seed=5, flags=abi.RT_RENDER_COUNTERS))
    rimg, rst = o.render(cam(w, h), orc.render_opts(w, h, spp, 8, seed=5, flags=abi.RT_RENDER_COUNTERS))
    assert img.tobytes() == rimg.tobytes() and st.segments == rst.segments
    assert st.nodes_visited == rst.nodes_visited and list(st.prims_tested) == list(rst.prims_tested)


def test_full_size_synthetic_scene_properties(scene_cache):
    _, cam, fr = scene_cache("synthetic")
    g = lib.Scene(fr, 4, abi.RT_SPLIT_HLBVH)
    info = g.info()
    assert info.n_prims == 2 * 708 * 708 + 10_000_000 and info.max_depth <= 64
    nodes, order = g.nodes(), g.prim_order()
    # leaf references beyond 2^23 and a deep tree are what this size adds
    assert info.n_nodes > (1 << 23) and info.max_depth >= 24
    dev = lib.Scene(fr, 4, abi.RT_SPLIT_HLBVH, build_on_device=True)
    assert dev.nodes().tobytes() == nodes.tobytes() and (dev.prim_order() == order).all()
    dev.close()
    o = orc.OracleScene(fr, 4, abi.RT_SPLIT_HLBVH)
    assert o.nodes().tobytes() == nodes.tobytes() and (o.prim_order() == order).all()
    del nodes, order
    w, h, spp, depth = 960, 540, 2, 8
    c = cam(w, h)
    a, sa = g.render(c, lib.render_opts(w, h, spp, depth, seed=0))
    b, sb = g.render(c, lib.render_opts(w, h, spp, depth, seed=0, max_paths=1 << 19))
    assert sb.batches == 2 and a.tobytes() == b.tobytes() and sa.segments == sb.segments
    assert np.isfinite(a).all() and (a[..., 3] == 1).all() and sa.segments > 2 * w * h * spp
    rng = np.random.default_rng(0)
    opts = orc.render_opts(w, h, spp, depth, seed=0)
    for pix in rng.integers(0, w * h, 64):
        rgb, _ = o.path_radiance(c, opts, int(pix) * spp, spp)
        acc = np.zeros(3, np.float32)
        for s in range(spp):
            acc = acc + rgb[s]
        assert (a[pix // w, pix % w, :3] == acc / np.float32(spp)).all(), pix
    # incoherent rays through both trace entry points against the oracle
    primary = lib.generate_rays(c, lib.render_opts(w, h, 1, seed=1), 0, 1 << 16)
    ref = o.intersect(primary, threads=16)
    assert_same_hits(g.intersect(primary), ref, "primary, full size")
    second = bounce_rays(primary, ref, 9)
    assert_same_hits(g.intersect(second), o.intersect(second, threads=16), "bounce 1, full size")
    assert (g.does_intersect(second) == o.occluded(second, threads=16)).all()


@pytest.mark.parametrize("cluster", ["7,8", "16", "3,4"])
def test_record_layout_does_not_change_results(monkeypatch, scene_cache, cluster):
    """The interior records of a tree in global memory may be laid out in clusters
    (RT355_NODE_CLUSTER = size[,alignment]) instead of the flattened tree's pre-order: record
    numbers are internal to the device tables, so hits, any-hit bytes, counters and frames stay
    those of the oracle, and rt_scene_get_nodes still returns the reference's order."""
    sd, cam = scenes.BUILDERS["synthetic"](grid=48, n_spheres=20000)
    fr = sd.freeze()
    monkeypatch.setenv("RT355_NODES", "global")
    monkeypatch.setenv("RT355_NODE_CLUSTER", cluster)
    g, o = lib.Scene(fr, 4, abi.RT_SPLIT_SAH), orc.OracleScene(fr, 4, abi.RT_SPLIT_SAH)
    assert g.nodes().tobytes() == o.nodes().tobytes()
    w, h = 192, 108
    c = cam(w, h)
    primary = lib.generate_rays(c, lib.render_opts(w, h, 1, seed=1), 0, w * h)
    ref, ref_counters = o.intersect(primary, counters=True)
    assert_same_hits(g.intersect(primary), ref, "clustered records, primary")
    second = bounce_rays(primary, ref, 3)
    assert_same_hits(g.intersect(second), o.intersect(second), "clustered records, bounce 1")
    assert (g.does_intersect(second) == o.occluded(second)).all()
    img, st = g.render(c, lib.render_opts(w, h, 2, 8, seed=2, flags=abi.RT_RENDER_COUNTERS))
    want, wst = o.render(c, orc.render_opts(w, h, 2, 8, seed=2, flags=abi.RT_RENDER_COUNTERS))
    assert img.tobytes() == want.tobytes() and st.segments == wst.segments
    assert st.nodes_visited == wst.nodes_visited and list(st.prims_tested) == list(wst.prims_tested)
