"""CPU tests of the product's host side: the C-ABI library loads and exports every
symbol include/rt355.h declares, struct layouts match the ctypes mirror, the host
BVH builder produces the oracle's tree bit for bit, argument errors come back as
codes (never aborts), and compute entry points refuse to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracer_amd import abi, lib
from raytracer_amd.scene import SceneData
from tests import orc
from tests.mixed import mixed_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "rt355.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_layouts_match():
    L = lib.load()
    out = (C.c_uint32 * 64)()
    n = L.rt_abi_sizes(out, 64)
    assert n == len(abi.ABI_SIZES)
    for (name, size), got in zip(abi.ABI_SIZES, out[:n]):
        assert size == got, (name, size, got)


@pytest.mark.parametrize("split", range(4))
@pytest.mark.parametrize("max_prims", [1, 4, 255, 1000])
def test_builder_matches_oracle_bit_for_bit(split, max_prims):
    fr = mixed_scene(seed=40 + split, n_each=80).freeze()
    g, o = lib.Scene(fr, max_prims, split), orc.OracleScene(fr, max_prims, split)
    assert g.nodes().tobytes() == o.nodes().tobytes()
    assert (g.prim_order() == o.prim_order()).all()
    assert (g.world_bound() == o.world_bound()).all()
    gi, oi = g.info(), o.info()
    assert (gi.n_prims, gi.n_nodes, gi.max_depth) == (oi.n_prims, oi.n_nodes, oi.max_depth)
    # max_prims_in_node is clamped to 255 (bvh.rs:114)
    assert g.nodes()["primitive_count"].max() <= 255 or split != abi.RT_SPLIT_HLBVH


@pytest.mark.parametrize("name", ["random", "cornell", "week", "pbrt"])
def test_named_scene_trees_match_oracle(scene_cache, name):
    fr = scene_cache(name)[2]
    for split in range(4):
        g, o = lib.Scene(fr, 4, split), orc.OracleScene(fr, 4, split)
        assert g.nodes().tobytes() == o.nodes().tobytes(), (name, split)
        assert (g.prim_order() == o.prim_order()).all()


def test_large_build_matches_oracle():
    s = SceneData()
    s.lambertian((1, 1, 1))
    rng = np.random.default_rng(3)
    s.add_sphere_block(rng.random((60000, 3)) * 100, rng.random(60000) * 0.3 + 0.05)
    tri = rng.random((40000, 1, 3)) * 100 + (rng.random((40000, 3, 3)) - 0.5)
    s.add_triangles(tri)
    fr = s.freeze()
    for split in (abi.RT_SPLIT_SAH, abi.RT_SPLIT_HLBVH):
        g, o = lib.Scene(fr, 4, split), orc.OracleScene(fr, 4, split)
        assert g.nodes().tobytes() == o.nodes().tobytes()
        assert (g.prim_order() == o.prim_order()).all()
        assert g.info().max_depth <= 64


def test_errors_are_codes_not_aborts():
    L = lib.load()
    h = C.c_void_p()
    opts = abi.rt_build_opts(4, 0)
    assert L.rt_scene_create(None, C.byref(opts), C.byref(h)) == abi.RT_ERR_INVALID
    assert b"null" in L.rt_last_error()
    # primitive index out of range
    s = SceneData()
    s.lambertian((1, 1, 1))
    s.add_sphere((0, 0, 0), 1.0)
    fr = s.freeze()
    fr.desc.n_spheres = 0
    assert L.rt_scene_create(C.byref(fr.desc), C.byref(opts), C.byref(h)) == abi.RT_ERR_INVALID
    assert b"malformed" in L.rt_last_error()
    with pytest.raises(lib.RtError):
        lib.Scene(fr)
    # unknown primitive kind, bad material, bad texture reference
    for mutate in ("kind", "material", "texture"):
        s = SceneData()
        s.lambertian((1, 1, 1))
        s.add_sphere((0, 0, 0), 1.0)
        fr = s.freeze()
        if mutate == "kind":
            fr._keep["prims"]["kind"][0] = 99
        elif mutate == "material":
            fr._keep["prims"]["material"][0] = 5
        else:
            fr._keep["materials"]["texture"][0] = 7
        assert L.rt_scene_create(C.byref(fr.desc), C.byref(opts), C.byref(h)) == abi.RT_ERR_INVALID, mutate
        assert orc.lib().orc_scene_create(C.byref(fr.desc), C.byref(opts), C.byref(h)) == abi.RT_ERR_INVALID
    assert L.rt_scene_world_bound(None, None) == abi.RT_ERR_INVALID
    L.rt_scene_destroy(None)  # no-op


def _one_sphere(noise=False):
    s = SceneData()
    if noise:
        s.set_perlin()
        s.lambertian(texture=s.add_texture(abi.RT_TEX_NOISE, scale=0.1))
    else:
        s.lambertian((1, 1, 1))
    s.add_sphere((0, 0, 0), 1.0)
    s.add_rect(1, 0.0, -1.0, 1.0, -1.0, 1.0)
    return s.freeze()


def test_descriptions_with_dangling_tables_or_bad_values_are_rejected():
    """Counts without tables, Perlin permutation entries the kernels would index out of
    bounds with, non-finite geometry and a rectangle axis above 2 all come back as
    RT_ERR_INVALID with a reason."""
    L = lib.load()
    h = C.c_void_p()
    opts = abi.rt_build_opts(4, 0)

    def create(fr):
        rc = L.rt_scene_create(C.byref(fr.desc), C.byref(opts), C.byref(h))
        return rc, L.rt_last_error().decode()

    fr = _one_sphere()
    assert create(fr)[0] == abi.RT_OK
    L.rt_scene_destroy(h)
    for field, word in (("materials", "materials is null"), ("textures", "textures is null"),
                        ("prims", "prims is null")):
        fr = _one_sphere()
        setattr(fr.desc, field, None)
        rc, msg = create(fr)
        assert rc == abi.RT_ERR_INVALID and word in msg, (field, msg)
    fr = _one_sphere()
    fr.desc.n_images = 1  # count without a table
    rc, msg = create(fr)
    assert rc == abi.RT_ERR_INVALID and "images is null" in msg
    for bad in (-1, 256, 1 << 20):
        fr = _one_sphere(noise=True)
        fr._keep["perlin"][1][2, 17] = bad
        rc, msg = create(fr)
        assert rc == abi.RT_ERR_INVALID and "perlin_perm" in msg, (bad, msg)
    for value in (np.nan, np.inf, -np.inf):
        fr = _one_sphere()
        fr._keep["spheres"]["c"][0, 1] = value
        rc, msg = create(fr)
        assert rc == abi.RT_ERR_INVALID and "finite" in msg, (value, msg)
    fr = _one_sphere()
    fr._keep["spheres"]["r"][0] = np.nan
    assert create(fr)[0] == abi.RT_ERR_INVALID
    fr = _one_sphere()
    fr._keep["rects"]["axis"][0] = 3
    rc, msg = create(fr)
    assert rc == abi.RT_ERR_INVALID and "axis" in msg


def test_sobol_tables_stop_at_the_reference_tables_last_row():
    """data/src/sobol.rs holds MATRICES / MATRICES_INV for resolutions up to 2^26: larger
    frames are RT_ERR_LIMIT, not an overrun of the 52-column tables."""
    assert lib.sobol_tables(1 << 26, 3)[1].shape == (52,)
    for w, h in (((1 << 26) + 1, 1), (5, 1 << 27), (0xFFFFFFFF, 0xFFFFFFFF), (0x80000001, 2)):
        with pytest.raises(lib.RtError) as e:
            lib.sobol_tables(w, h)
        assert e.value.code == abi.RT_ERR_LIMIT
    # the same bound guards the entry points that take a sampler (checked before any device use)
    import torch
    if not torch.cuda.is_available():
        cam = abi.rt_camera()
        opts = lib.render_opts((1 << 26) + 1, 1, 1, sampler=abi.RT_SAMPLER_SOBOL)
        rays = np.zeros(1, dtype=abi.ray_dtype)
        rc = lib.load().rt_generate_rays(C.byref(cam), C.byref(opts), 0, 1, rays.ctypes.data, 0)
        assert rc in (abi.RT_ERR_LIMIT, abi.RT_ERR_NO_DEVICE)


def test_no_alloc_entry_point_argument_errors():
    """rt_intersect_dev_noalloc never uploads (that would allocate): a scene that is not
    resident, a batch above the fixed maximum and null buffers are errors."""
    L = lib.load()
    g = lib.Scene(_one_sphere())
    assert L.rt_intersect_dev_noalloc(None, None, 0, None, 0, None) == abi.RT_ERR_INVALID
    rc = L.rt_intersect_dev_noalloc(g._h, 16, abi.RT_NOALLOC_MAX_RAYS + 1, 16, 0, None)
    assert rc == abi.RT_ERR_LIMIT and b"at most" in L.rt_last_error()
    assert L.rt_intersect_dev_noalloc(g._h, None, 4, None, 0, None) == abi.RT_ERR_INVALID
    rc = L.rt_intersect_dev_noalloc(g._h, 16, 4, 16, 0, None)  # never uploaded
    assert rc == abi.RT_ERR_INVALID and b"rt_scene_upload" in L.rt_last_error()
    header = open(os.path.join(ROOT, "include", "rt355.h")).read()
    assert "#define RT_NOALLOC_MAX_RAYS (1u << 28)" in header and abi.RT_NOALLOC_MAX_RAYS == 1 << 28


def test_deep_tree_is_rejected_like_the_reference_stack():
    # centroids spaced geometrically make the middle split peel one primitive per level:
    # the tree needs more than the 64 stack entries of bvh.rs:753
    s = SceneData()
    s.lambertian((1, 1, 1))
    for k in range(80):
        c = 4.0 ** (k - 40)
        s.add_sphere((c, 0.0, 0.0), 0.1 * c)
    fr = s.freeze()
    with pytest.raises(lib.RtError) as e:
        lib.Scene(fr, 1, abi.RT_SPLIT_MIDDLE)
    assert e.value.code == abi.RT_ERR_DEPTH
    with pytest.raises(orc.OracleError):
        orc.OracleScene(fr, 1, abi.RT_SPLIT_MIDDLE)
    assert lib.Scene(fr, 1, abi.RT_SPLIT_EQUAL_COUNTS).info().max_depth <= 9


def test_compute_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    s = SceneData()
    s.lambertian((1, 1, 1))
    s.add_sphere((0, 0, 0), 1.0)
    g = lib.Scene(s.freeze())
    rays = np.zeros(4, dtype=abi.ray_dtype)
    for call in (lambda: g.intersect(rays), lambda: g.does_intersect(rays), lambda: g.upload()):
        with pytest.raises(lib.RtError) as e:
            call()
        assert e.value.code == abi.RT_ERR_NO_DEVICE and "no CPU path" in str(e.value)
    # a negative device index is an error as well, never a CPU fallback
    g.device = -1
    with pytest.raises(lib.RtError):
        g.intersect(rays)


def test_missing_library_is_an_import_error(monkeypatch):
    monkeypatch.setattr(lib, "_LIB", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/librt355.so")
    with pytest.raises(ImportError):
        lib.load()


def test_media_that_enclose_the_scene_are_kept_out_of_the_tree(scene_cache):
    """A medium whose sphere contains the box around every other primitive is not given to
    the tree (every ray would meet it there): the library and the oracle pick the same ones,
    the tree is the tree of the scene without them, and the world bound still covers them."""
    _, _, fr = scene_cache("week")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    gi, oi = g.info(), o.info()
    prims = fr._keep["prims"]
    assert gi.n_enclosing == oi.n_enclosing == 1 and gi.enclosing[0] == oi.enclosing[0]
    fog = prims[gi.enclosing[0]]
    assert fog["kind"] & 0xFF == abi.RT_MEDIUM
    assert fr._keep["spheres"][fr._keep["media"][fog["index"]]["boundary_index"]]["r"] == 5000.0
    assert len(g.prim_order()) == gi.n_prims - 1 and gi.enclosing[0] not in set(g.prim_order())
    assert list(g.world_bound()) == list(o.world_bound()) == [-5000.0] * 3 + [5000.0] * 3
    root = g.nodes()[0]
    assert root["hi"][1] < 1000.0  # the tree's own root is the size of the scene, not of the fog

    def build(n_around, r_small=3.0, with_solid=True):
        s = SceneData()
        s.lambertian((1, 1, 1))
        iso = s.isotropic((1, 1, 1))
        if with_solid:
            for k in range(20):
                s.add_sphere((k - 10.0, 0.5 * k, 2.0), 0.7)
        s.add_medium(0.3, iso, sphere=((0.0, 0.0, 0.0), r_small))  # inside the scene: stays in the tree
        for k in range(n_around):
            s.add_medium(0.01, iso, sphere=((0.1 * k, 0.0, 0.0), 100.0 + k))
        return s.freeze()

    for n_around, expect in ((0, 0), (1, 1), (4, 4), (6, 4)):
        fr2 = build(n_around)
        for split in range(4):
            g2, o2 = lib.Scene(fr2, 2, split), orc.OracleScene(fr2, 2, split)
            i2 = g2.info()
            assert i2.n_enclosing == o2.info().n_enclosing == expect
            assert list(i2.enclosing[:expect]) == list(range(21, 21 + expect))
            assert g2.nodes().tobytes() == o2.nodes().tobytes() and (g2.prim_order() == o2.prim_order()).all()
            assert (g2.world_bound() == o2.world_bound()).all()
        # the tree is the one of the same scene without the media kept out of it
        if n_around == 4:
            assert lib.Scene(fr2).nodes().tobytes() == lib.Scene(build(0)).nodes().tobytes()
    # a scene of media only: nothing to enclose, nothing is kept out
    fr3 = build(2, with_solid=False)
    assert lib.Scene(fr3).info().n_enclosing == orc.OracleScene(fr3).info().n_enclosing == 0
    # a sphere that cuts through the scene's box is not enclosing
    fr4 = build(0, r_small=9.0)
    assert lib.Scene(fr4).info().n_enclosing == 0
