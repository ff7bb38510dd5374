"""CPU tests of the oracle: the reference's own known answers that touch the path
(AABB union / distance, geometry/src/tests/bounds.rs), known-answer vectors for the
slab, sphere and triangle tests, the published PCG32 vector, BVH == brute force for
all four builders, and an independent fp64 brute-force cross-check."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, pcg32
from raytracer_amd.scene import SceneData, rotate_y, translate
from tests import orc
from tests.conftest import random_rays
from tests.mixed import mixed_scene

F = np.float32
BIG = np.finfo(np.float32).max


def f6(*v):
    return (C.c_float * 6)(*v)


def f3(*v):
    return (C.c_float * 3)(*v)


def ray(o, d, t_max=np.inf, time=0.0):
    r = np.zeros(1, dtype=abi.ray_dtype)
    r["o"], r["d"], r["t_max"], r["time"] = o, d, t_max, time
    return r


# ---- reference known answers: geometry/src/tests/bounds.rs ---------------------------------
def test_union3_reference_vectors():
    L = orc.lib()
    a = f6(-10, -10, 5, 0, 20, 10)
    zero = f6(BIG, BIG, BIG, -BIG, -BIG, -BIG)  # Bounds3::ZERO
    out = f6()
    L.orc_bounds_union(a, zero, out)
    assert list(out) == list(a)  # bounds.rs:102-105
    L.orc_bounds_union(zero, zero, out)
    assert list(out) == list(zero)  # bounds.rs:106
    d = f6(-15, 10, 30, -15, 10, 30)  # at_point
    L.orc_bounds_union(a, d, out)
    assert list(out) == [-15, -10, 5, 0, 20, 30]  # bounds.rs:108-116


def test_point_distance_reference_vectors():
    L = orc.lib()
    b = f6(0, 0, 0, 1, 1, 1)

    def d2(p):
        return L.orc_bounds_distance_squared(b, f3(*p))

    for p in [(0.5, 0.5, 0.5), (0, 1, 1), (0.25, 0.8, 1), (0, 0.25, 0.8), (0.7, 0, 0.8)]:
        assert d2(p) == 0.0  # bounds.rs:42-46
    assert np.sqrt(F(d2((6, 1, 1)))) == 5.0 and np.sqrt(F(d2((0, -10, 1)))) == 10.0  # :49-50
    for p, e in [((0.5, 0.5, 3), 2), ((0.5, 0.5, -3), 3), ((0.5, 3, 0.5), 2), ((0.5, -3, 0.5), 3),
                 ((3, 0.5, 0.5), 2), ((-3, 0.5, 0.5), 3)]:
        assert np.sqrt(F(d2(p))) == e  # :53-58
    assert d2((4, 8, -10)) == 3 * 3 + 7 * 7 + 10 * 10  # :61-64
    assert d2((-6, -10, 8)) == 6 * 6 + 10 * 10 + 7 * 7  # :65-68
    b = f6(-1, -3, 5, 2, -2, 18)
    assert L.orc_bounds_distance_squared(b, f3(-0.99, -2, 5)) == 0.0  # :72
    assert L.orc_bounds_distance_squared(b, f3(-3, -9, 22)) == 2 * 2 + 6 * 6 + 4 * 4  # :73-76


def test_surface_area_is_the_corrected_formula():
    assert orc.lib().orc_bounds_surface_area(f6(0, 0, 0, 1, 2, 3)) == 2 * (1 * 2 + 1 * 3 + 2 * 3)


# ---- PCG32 -------------------------------------------------------------------------------------
def pcg32_srandom(initstate, initseq, n):
    """The published PCG32 reference generator (pcg_basic): used to pin the LCG
    constants and the output permutation with its demo vector."""
    M = (1 << 64) - 1
    state, inc = 0, ((initseq << 1) | 1) & M
    out = []

    def step():
        nonlocal state
        old = state
        state = (old * 6364136223846793005 + inc) & M
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((x >> rot) | (x << ((-rot) & 31))) & 0xFFFFFFFF

    step()
    state = (state + initstate) & M
    step()
    for _ in range(n):
        out.append(step())
    return out


def test_pcg32_published_vector_and_streams():
    assert pcg32_srandom(42, 54, 6) == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B,
                                        0xCBED606E]
    for seq in (0, 1, 7, 2 ** 40 + 3):
        want = pcg32_srandom(0x853C49E6748FEA9B, seq, 64)  # Rng::new(seq), rng.rs:26-39
        got = np.zeros(64, np.uint32)
        orc.lib().orc_pcg32_stream(seq, got.ctypes.data, 64)
        assert list(got) == want
        r = pcg32.Rng(seq)
        assert [r.uniform_u32() for _ in range(64)] == want
    fl = np.zeros(1000, np.float32)
    orc.lib().orc_pcg32_floats(5, fl.ctypes.data, 1000)
    r = pcg32.Rng(5)
    assert [r.uniform_float() for _ in range(1000)] == list(map(float, fl))
    assert (pcg32.streams(5, 3, 10)[0] == fl[:10]).all()
    assert fl.min() >= 0.0 and fl.max() <= 1.0 - 2.0 ** -23


def test_math_polynomials_are_accurate():
    L = orc.lib()
    xs = np.linspace(-50, 50, 4001).astype(F)
    assert max(abs(L.orc_math_eval(0, float(x), 0) - np.sin(np.float64(x))) for x in xs) < 3e-6
    assert max(abs(L.orc_math_eval(1, float(x), 0) - np.cos(np.float64(x))) for x in xs) < 3e-6
    xs = np.exp(np.linspace(-17, 0, 2000)).astype(F)
    assert max(abs(L.orc_math_eval(2, float(x), 0) / np.log(np.float64(x)) - 1) for x in xs[:-1]) < 3e-7
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(4000, 2)).astype(F)
    assert max(abs(L.orc_math_eval(3, float(y), float(x)) - np.arctan2(np.float64(y), np.float64(x)))
               for y, x in pts) < 5e-7
    xs = np.linspace(-1, 1, 2001).astype(F)
    assert max(abs(L.orc_math_eval(4, float(x), 0) - np.arccos(np.float64(x))) for x in xs) < 6e-4
    assert max(abs(L.orc_math_eval(4, float(x), 0) - np.arccos(np.float64(x))) for x in xs[10:-10]) < 3e-6
    eps = 2.0 ** -23  # gamma(n), geometry/src/float.rs:44-46 with f32::EPSILON
    for n in (2, 3, 5, 7):
        assert L.orc_math_eval(5, float(n), 0) == F(F(n * eps) / F(1 - F(n * eps)))


def test_offset_ray_origin_moves_outward_by_at_least_the_error():
    L = orc.lib()
    rng = np.random.default_rng(1)
    for _ in range(200):
        p = (rng.normal(size=3) * 100).astype(F)
        e = (np.abs(rng.normal(size=3)) * 1e-4).astype(F)
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(F)
        w = rng.normal(size=3).astype(F)
        out = np.zeros(3, F)
        L.orc_offset_ray_origin(p.ctypes.data, e.ctypes.data, n.ctypes.data, w.ctypes.data, out.ctypes.data)
        side = 1.0 if np.dot(n, w) >= 0 else -1.0
        moved = np.dot(out.astype(np.float64) - p, n) * side
        assert moved >= np.dot(np.abs(n), e) * 0.999


# ---- known answers for the intersection routines -----------------------------------------------------
def test_slab_known_answers():
    L = orc.lib()
    box = f6(0, 0, 0, 1, 1, 1)

    def hit(o, d, t_max=np.inf):
        r = ray(o, d, t_max)
        return L.orc_slab_test(box, r.ctypes.data)

    assert hit((0.5, 0.5, -1), (0, 0, 1)) == 1
    assert hit((0.5, 0.5, -1), (0, 0, -1)) == 0  # box behind the ray
    assert hit((0.5, 0.5, -1), (0, 0, 1), t_max=0.5) == 0  # bounds3.rs:237 t_min < t_max
    assert hit((0.5, 0.5, -1), (0, 0, 1), t_max=1.5) == 1
    assert hit((0.5, 0.5, 0.5), (1, 0, 0)) == 1  # origin inside
    assert hit((2, 0.5, 0.5), (0, 1, 0)) == 0  # parallel, outside the x slab: 1/0 path
    assert hit((0.5, 0.5, 0.5), (0, 1, 0)) == 1  # parallel, inside
    assert hit((-1, -1, -1), (1, 1, 1)) == 1  # through both corners
    assert hit((-1, 2.0001, 0.5), (1, 0, 0)) == 0


def single(scene_fn):
    s = SceneData()
    s.lambertian((1, 1, 1))
    scene_fn(s)
    return orc.OracleScene(s.freeze())


def test_sphere_known_answers():
    o = single(lambda s: s.add_sphere((0, 0, 0), 1.0))
    h = o.intersect(ray((0, -5, 0), (0, 1, 0)))[0]
    assert h["prim"] == 0 and h["t"] == 4.0 and tuple(h["n"]) == (0, -1, 0)
    h = o.intersect(ray((0, 0, 0), (0, 1, 0)))[0]  # from inside: far root
    assert h["t"] == 1.0 and tuple(h["n"]) == (0, 1, 0)
    h = o.intersect(ray((0, 0, -5), (0, 0, 1)))[0]  # at the pole x is nudged to 1e-5 r (sphere.rs:72-74)
    assert h["t"] == 4.0 and tuple(h["n"]) == (np.float32(1e-5), 0, -1)
    assert o.intersect(ray((0, 0, -5), (0, 0, 1), t_max=3.9))[0]["prim"] == abi.RT_MISS
    # t == t_max: the root box is entered exactly at t_max and the slab test is strict
    # (bounds3.rs:237), so the tree culls it ...
    assert o.intersect(ray((0, 0, -5), (0, 0, 1), t_max=4.0))[0]["prim"] == abi.RT_MISS
    # ... while a hit at t == t_max inside the box is accepted (quadric.rs:24)
    t = o.intersect(ray((0.5, -5, 0), (0, 1, 0)))[0]["t"]
    assert o.intersect(ray((0.5, -5, 0), (0, 1, 0), t_max=t))[0]["prim"] == 0
    assert o.intersect(ray((0, 2, -5), (0, 0, 1)))[0]["prim"] == abi.RT_MISS
    assert o.intersect(ray((0, 0, 5), (0, 0, 1)))[0]["prim"] == abi.RT_MISS  # behind
    h = o.intersect(ray((-5, 0, 0), (1, 0, 0)))[0]  # u = phi / 2pi with phi = pi, v = 0.5
    assert abs(h["u"] - 0.5) < 1e-6 and abs(h["v"] - 0.5) < 1e-6
    # negative radius flips the normal
    o = single(lambda s: s.add_sphere((0, 0, 0), -1.0))
    assert tuple(o.intersect(ray((0, -5, 0), (0, 1, 0)))[0]["n"]) == (0, 1, 0)
    # moving sphere: centre follows ray.time
    o = single(lambda s: s.add_moving_sphere((0, 0, 0), (2, 0, 0), 0.0, 1.0, 1.0))
    assert o.intersect(ray((0, 0, -5), (0, 0, 1), time=0.0))[0]["t"] == 4.0
    assert o.intersect(ray((2, 0, -5), (0, 0, 1), time=1.0))[0]["t"] == 4.0
    assert o.intersect(ray((2, 0, -5), (0, 0, 1), time=0.0))[0]["prim"] == abi.RT_MISS


def test_general_sphere_matches_translated_sphere_and_clips():
    # full general sphere == fast sphere up to rounding (1e-5 relative bar)
    fast = single(lambda s: s.add_sphere((3, -2, 5), 1.5))
    gen = single(lambda s: s.add_gsphere(translate((3, -2, 5)), 1.5))
    rays = random_rays(20000, (0, -5, 2), (6, 1, 8), 3)
    a, b = fast.intersect(rays), gen.intersect(rays)
    both = (a["prim"] == 0) & (b["prim"] == 0)
    assert both.sum() > 1000 and (a["prim"] != b["prim"]).mean() < 2e-3  # grazing rays only
    assert np.abs(a["t"][both] - b["t"][both]).max() <= 2e-5 * np.abs(a["t"][both]).max()
    assert np.abs(a["n"][both] - b["n"][both]).max() < 2e-4
    assert np.abs(a["v"][both] - b["v"][both]).max() < 1e-4
    # z clipping (sphere.rs:82-108): a ray down the pole passes through the removed cap
    cut = single(lambda s: s.add_gsphere(np.eye(4), 1.0, z=(-0.5, 0.5)))
    assert cut.intersect(ray((0, 0, -5), (0, 0, 1)))[0]["prim"] == abi.RT_MISS
    h = cut.intersect(ray((-5, 0, 0.2), (1, 0, 0)))[0]
    assert h["prim"] == 0 and abs(h["t"] - (5 - np.sqrt(1 - 0.04))) < 1e-5
    # phi clipping: phi_max = 180 keeps y >= 0; a ray entering through y < 0 hits the far side
    half = single(lambda s: s.add_gsphere(np.eye(4), 1.0, phi_max=180.0))
    h = half.intersect(ray((0, -5, 0), (0, 1, 0)))[0]
    assert h["prim"] == 0 and abs(h["t"] - 6.0) < 1e-5
    # reverse_orientation ^ swaps_handedness (shapes/src/lib.rs:68-70)
    rev = single(lambda s: s.add_gsphere(np.eye(4), 1.0, reverse=True))
    assert rev.intersect(ray((0, 0, -5), (0, 0, 1)))[0]["n"][2] > 0.99
    mir = single(lambda s: s.add_gsphere(np.diag([1.0, 1.0, -1.0, 1.0]), 1.0))
    assert mir.intersect(ray((0, 0, -5), (0, 0, 1)))[0]["n"][2] > 0.99


def test_triangle_known_answers_and_watertight_edge():
    tri = [[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]
    o = single(lambda s: s.add_triangles(tri))
    h = o.intersect(ray((0.25, 0.25, 1), (0, 0, -1)))[0]
    assert h["prim"] == 0 and h["t"] == 1.0
    assert abs(h["u"] - 0.25) < 1e-7 and abs(h["v"] - 0.25) < 1e-7  # default uvs, triangle.rs:72-78
    assert tuple(abs(h["n"])) == (0, 0, 1)
    assert o.intersect(ray((0.75, 0.75, 1), (0, 0, -1)))[0]["prim"] == abi.RT_MISS
    assert o.intersect(ray((0.25, 0.25, -1), (0, 0, -1)))[0]["prim"] == abi.RT_MISS
    # two triangles sharing the diagonal: rays aimed exactly at the shared edge hit exactly one
    quad = [[[-1, -1, 0], [1, -1, 0], [1, 1, 0]], [[-1, -1, 0], [1, 1, 0], [-1, 1, 0]]]
    o = single(lambda s: s.add_triangles(quad))
    s = np.linspace(-0.99, 0.99, 199, dtype=F)
    rays = np.zeros(len(s), dtype=abi.ray_dtype)
    rays["o"] = np.stack([s, s, np.full_like(s, 3)], axis=1)
    rays["d"] = (0, 0, -1)
    rays["t_max"] = np.inf
    h = o.intersect(rays)
    assert (h["prim"] != abi.RT_MISS).all() and (np.abs(h["t"] - 3.0) < 1e-6).all()
    # degenerate triangle never hits
    o = single(lambda s: s.add_triangles([[[0, 0, 0], [1, 1, 1], [2, 2, 2]]]))
    assert o.intersect(ray((0.5, 0.5, 5), (0, 0, -1)))[0]["prim"] == abi.RT_MISS


def test_rect_box_medium_known_answers():
    o = single(lambda s: s.add_rect(1, 2.0, -1, 1, -1, 1))  # XzRect at y = 2
    h = o.intersect(ray((0.5, 0, 0.5), (0, 1, 0)))[0]
    assert h["prim"] == 0 and h["t"] == 2.0 and tuple(h["n"]) == (0, 1, 0)
    assert (h["u"], h["v"]) == (0.75, 0.75) and (h["flags"] >> 8) == 1
    assert o.intersect(ray((1.5, 0, 0.5), (0, 1, 0)))[0]["prim"] == abi.RT_MISS
    assert o.intersect(ray((0.5, 0, 0.5), (1, 0, 0)))[0]["prim"] == abi.RT_MISS  # parallel
    o = single(lambda s: s.add_box((0, 0, 0), (1, 2, 3)))
    h = o.intersect(ray((0.5, 1, -4), (0, 0, 1)))[0]
    assert h["t"] == 4.0 and tuple(h["n"]) == (0, 0, -1) and (h["flags"] >> 8) == 4
    h = o.intersect(ray((0.5, 1, 1), (0, 0, 1)))[0]  # from inside: exit face
    assert h["t"] == 2.0 and tuple(h["n"]) == (0, 0, 1) and (h["flags"] >> 8) == 5
    # rotated box: same hit distance as rotating the ray the other way
    m = translate((5, 0, 0)) @ rotate_y(30.0)
    o = single(lambda s: s.add_box((-1, -1, -1), (1, 1, 1), xform=s.add_xform(m)))
    h = o.intersect(ray((5, 0, -10), (0, 0, 1)))[0]
    want = 10 - 1 / np.cos(np.radians(30))
    assert h["prim"] == 0 and abs(h["t"] - want) < 1e-4
    # medium: hit distance always inside the boundary, about 1 - exp(-density * chord) of rays scatter
    o = single(lambda s: s.add_medium(0.5, 0, sphere=((0, 0, 0), 1.0)))
    rng = np.random.default_rng(0)
    n = 20000
    rays = np.zeros(n, dtype=abi.ray_dtype)
    rays["o"] = np.stack([rng.random(n) * 1e-3, rng.random(n) * 1e-3, np.full(n, -5.0)], axis=1)
    rays["d"] = (0, 0, 1)
    rays["t_max"] = np.inf
    h = o.intersect(rays)
    hit = h["prim"] == 0
    assert abs(hit.mean() - (1 - np.exp(-0.5 * 2))) < 0.02
    assert (h["t"][hit] > 4.0).all() and (h["t"][hit] < 6.0).all()
    # scatter depths follow the exponential law: recover the uniform variate and test it for
    # flatness and for independence between rays that differ in the last bits of one coordinate
    rays["o"] = np.stack([np.float32(0.1) + np.arange(n, dtype=np.float32) * np.float32(2.0 ** -20),
                          np.zeros(n, np.float32), np.full(n, -5.0, np.float32)], axis=1)
    o2 = single(lambda s: s.add_medium(50.0, 0, sphere=((0, 0, 0), 1.0)))  # dense: every ray scatters
    h = o2.intersect(rays)
    assert (h["prim"] == 0).all()
    depth = h["t"].astype(np.float64) - (5.0 - np.sqrt(1.0 - rays["o"][:, 0].astype(np.float64) ** 2))
    u = np.exp(-50.0 * depth)
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1.0 / 12.0) < 0.005
    assert (np.abs(np.histogram(u, bins=20, range=(0, 1))[0] / n - 0.05) < 0.01).all()
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.03 and abs(np.corrcoef(u[:-2], u[2:])[0, 1]) < 0.03


# ---- structure ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("split", range(4))
def test_bvh_equals_brute_force(split):
    fr = mixed_scene(seed=20 + split, n_each=50).freeze()
    o = orc.OracleScene(fr, 4, split)
    wb = o.world_bound()
    rays = random_rays(30000, np.maximum(wb[:3], -30), np.minimum(wb[3:], 30), split)
    a, b = o.intersect(rays), o.intersect_brute(rays)
    # the tree only prunes: same closest hit (ties excepted, none expected in random scenes)
    assert (a["prim"] == b["prim"]).all() and (a["t"] == b["t"]).all()
    assert ((a["prim"] != abi.RT_MISS) == (o.occluded(rays) != 0)).all()
    nodes, order = o.nodes(), o.prim_order()
    info = o.info()
    # every primitive is in exactly one leaf, except the medium around the whole scene, which
    # is kept out of the tree and tested before the traversal
    kept_out = list(info.enclosing[:info.n_enclosing])
    assert len(kept_out) == 1 and (o.frozen._keep["prims"]["kind"][kept_out[0]] & 0xFF) == abi.RT_MEDIUM
    assert sorted(list(order) + kept_out) == list(range(info.n_prims))
    leaves = nodes[nodes["primitive_count"] > 0]
    assert leaves["primitive_count"].sum() == len(order)
    inner = np.nonzero(nodes["primitive_count"] == 0)[0]
    assert (nodes["child_offset"][inner] > inner + 1).all() and (nodes["axis"] < 3).all()
    assert o.info().max_depth <= 64


def test_empty_and_single_primitive_scenes():
    o = orc.OracleScene(SceneData().freeze())
    assert o.info().n_nodes == 0 and (o.world_bound()[:3] > o.world_bound()[3:]).all()
    rays = random_rays(10, (-1, -1, -1), (1, 1, 1), 0)
    assert (o.intersect(rays)["prim"] == abi.RT_MISS).all() and not o.occluded(rays).any()
    o = single(lambda s: s.add_sphere((0, 0, 0), 1.0))
    assert o.info().n_nodes == 1 and list(o.world_bound()) == [-1, -1, -1, 1, 1, 1]


def test_coincident_centroids_make_one_leaf():
    # 300 identical spheres: no split possible, one leaf (bvh.rs:181-191); count fits u16
    def build(s):
        for _ in range(300):
            s.add_sphere((1, 2, 3), 0.5)

    o = single(build)
    nodes = o.nodes()
    assert len(nodes) == 1 and nodes[0]["primitive_count"] == 300
    h = o.intersect(ray((1, 2, -5), (0, 0, 1)))[0]
    assert h["t"] == 7.5 and h["prim"] == 299  # ties: the primitive tested last wins


def test_fp64_brute_force_cross_check():
    """Independent float64 intersector (numpy) for spheres, triangles, rects and boxes:
    same hit/miss and primitive except where the fp64 margin is tiny (grazing rays),
    t within 1e-5 relative."""
    rng = np.random.default_rng(7)
    s = SceneData()
    s.lambertian((1, 1, 1))
    sph = [(rng.random(3) * 20 - 10, rng.random() * 1.5 + 0.3) for _ in range(40)]
    tri = rng.random((40, 1, 3)) * 20 - 10 + (rng.random((40, 3, 3)) - 0.5) * 4
    rects = [(int(rng.integers(3)), rng.random() * 20 - 10, *sorted(rng.random(2) * 20 - 10),
              *sorted(rng.random(2) * 20 - 10)) for _ in range(10)]
    boxes = [(lo := rng.random(3) * 18 - 10, lo + rng.random(3) * 2 + 0.2) for _ in range(10)]
    for c, r in sph:
        s.add_sphere(c, r)
    s.add_triangles(tri)
    for r in rects:
        s.add_rect(*r)
    for lo, hi in boxes:
        s.add_box(lo, hi)
    o = orc.OracleScene(s.freeze())
    rays = random_rays(6000, (-10, -10, -10), (10, 10, 10), 9, finite_fraction=0.0, axis_fraction=0.0)
    got = o.intersect(rays)
    n = len(rays)

    def closest64(O, D):
        best_t = np.full(n, np.inf)
        best_p = np.full(n, -1)

        def consider(t, ok, pid):
            nonlocal best_t, best_p
            t = np.where(ok & (t > 0), t, np.inf)
            better = t < best_t
            best_p = np.where(better, pid, best_p)
            best_t = np.where(better, t, best_t)

        pid = 0
        with np.errstate(all="ignore"):
            for c, r in sph:
                c = np.asarray(np.float32(c), np.float64)
                r = float(np.float32(r))
                oc = O - c
                a = (D * D).sum(1)
                b = (oc * D).sum(1)
                disc = b * b - a * ((oc * oc).sum(1) - r * r)
                sq = np.sqrt(np.maximum(disc, 0))
                t0, t1 = (-b - sq) / a, (-b + sq) / a
                consider(np.where(t0 > 0, t0, t1), disc >= 0, pid)
                pid += 1
            for v in np.float32(tri).astype(np.float64):  # Moller-Trumbore
                e1, e2 = v[1] - v[0], v[2] - v[0]
                p = np.cross(D, e2)
                det = p @ e1
                tv = O - v[0]
                u = (tv * p).sum(1) / det
                q = np.cross(tv, e1)
                w = (D * q).sum(1) / det
                consider((q @ e2) / det, (u >= 0) & (w >= 0) & (u + w <= 1) & (np.abs(det) > 1e-12), pid)
                pid += 1
            for axis, k, a0, a1, b0, b1 in rects:
                A, B = [(1, 2), (0, 2), (0, 1)][axis]
                k, a0, a1, b0, b1 = (float(np.float32(x)) for x in (k, a0, a1, b0, b1))
                t = (k - O[:, axis]) / D[:, axis]
                a, b = O[:, A] + t * D[:, A], O[:, B] + t * D[:, B]
                consider(t, (a >= a0) & (a <= a1) & (b >= b0) & (b <= b1), pid)
                pid += 1
            for lo, hi in boxes:
                lo, hi = np.float32(lo).astype(np.float64), np.float32(hi).astype(np.float64)
                t0, t1 = (lo - O) / D, (hi - O) / D
                tn, tf = np.minimum(t0, t1).max(1), np.maximum(t0, t1).min(1)
                consider(np.where(tn > 0, tn, tf), tn <= tf, pid)
                pid += 1
        return best_t, best_p

    O, D = rays["o"].astype(np.float64), rays["d"].astype(np.float64)
    best_t, best_p = closest64(O, D)
    # a ray is "clear" (not grazing) when small perturbations of it keep the same winner
    clear = np.ones(n, bool)
    prng = np.random.default_rng(1)
    for _ in range(4):
        _, p2 = closest64(O + prng.normal(size=O.shape) * 1e-4, D + prng.normal(size=D.shape) * 1e-5)
        clear &= p2 == best_p
    assert clear.mean() > 0.9
    got_prim = got["prim"].astype(np.int64)
    got_prim[got["prim"] == abi.RT_MISS] = -1
    assert (got_prim[clear] == best_p[clear]).all()
    h = clear & (best_p >= 0)
    assert h.sum() > 1000
    assert (np.abs(got["t"][h] - best_t[h]) <= 1e-5 * best_t[h]).all()


def test_scene_builders(scene_cache):
    assert scene_cache("random")[0].n_prims() in range(470, 489)
    assert scene_cache("cornell")[0].n_prims() == 8
    assert scene_cache("week")[0].n_prims() == 400 + 1 + 1 + 3 + 2 + 1 + 1 + 1000
    assert scene_cache("pbrt")[0].n_prims() == 3  # sphere + indices.len() / 3 triangles


def test_render_is_deterministic_and_thread_independent(scene_cache):
    sd, cam, fr = scene_cache("cornell")
    o = orc.OracleScene(fr)
    c = cam(48, 48)
    a, sa = o.render(c, orc.render_opts(48, 48, 4, 30, seed=2), threads=1)
    b, sb = o.render(c, orc.render_opts(48, 48, 4, 30, seed=2), threads=7)
    assert a.tobytes() == b.tobytes() and sa.segments == sb.segments
    d, _ = o.render(c, orc.render_opts(48, 48, 4, 30, seed=3), threads=7)
    assert a.tobytes() != d.tobytes()
    assert np.isfinite(a).all() and a[..., :3].mean() > 0.05
    # tile shards add up to the frame
    tot = np.zeros_like(a)
    for k in range(2):
        part, _ = o.render(c, orc.render_opts(48, 48, 4, 30, seed=2, shard=(k, 2)), threads=4)
        tot += part
    assert tot.tobytes() == a.tobytes()
