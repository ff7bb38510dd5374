"""The HIP path's output against the independent float64 intersector (tests/fp64_ref.py),
through the C ABI and without the oracle anywhere in the comparison: hit / miss and primitive
index equal on every ray that is not grazing, t within 1e-5 relative, geometric normals and
(u, v) within the bounds stated in tests/fp64_ref.py -- on seeded scenes holding every surface
kind, on the reference's clipped, transformed spheres, and on camera and random rays of the
four named benchmark scenes.  Media are part of it: the float64 intersector restates the hash
behind a medium's free-flight distance bit for bit and everything after it in float64
(tests/fp64_ref.py), so scatter points inside fog are checked like surface hits."""
import numpy as np
import pytest

from raytracer_amd import abi, lib
from tests import fp64_ref
from tests.conftest import random_rays, set_node_source
from tests.mixed import mixed_scene
from tests.test_fp64_cpu import TOL, check_against_fp64, clipped_spheres_scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_hip_matches_fp64_on_every_surface_kind(seed):
    fr = mixed_scene(seed=seed, n_each=40, with_media=False).freeze()
    g = lib.Scene(fr)
    rays = random_rays(20000, (-12, -12, -12), (12, 12, 12), 70 + seed, finite_fraction=0.2, axis_fraction=0.0)
    out, best, clear = check_against_fp64(g.intersect, fr, rays, min_hits=3000)
    kinds = set(np.unique(fr._keep["prims"]["kind"][best["prim"][clear & (best["prim"] >= 0)]] & 0xFF))
    assert kinds == set(range(6)), kinds
    # any-hit agrees with the float64 closest hit on the same rays
    assert ((g.does_intersect(rays) != 0) == (best["prim"] >= 0))[clear].all()


@pytest.mark.parametrize("seed", [4, 5])
def test_hip_matches_fp64_with_media(seed):
    """The mixed scene with its three media: fog in a sphere, smoke in a rotated box, thin fog
    around everything (kept out of the tree and tested by the traversal's first turn)."""
    fr = mixed_scene(seed=seed, n_each=40).freeze()
    g = lib.Scene(fr)
    assert g.info().n_enclosing == 1
    rays = random_rays(20000, (-12, -12, -12), (12, 12, 12), 80 + seed, finite_fraction=0.2, axis_fraction=0.0)
    out, best, clear = check_against_fp64(g.intersect, fr, rays, min_clear=0.8, min_hits=3000)
    kinds = fr._keep["prims"]["kind"] & 0xFF
    hit_prims = best["prim"][clear & (best["prim"] >= 0)]
    media = np.nonzero(kinds == abi.RT_MEDIUM)[0]
    per_medium = [(hit_prims == m).sum() for m in media]
    assert len(media) == 3 and min(per_medium) > 50, per_medium
    assert ((g.does_intersect(rays) != 0) == (best["prim"] >= 0))[clear].all()


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_hip_matches_fp64_on_clipped_transformed_spheres(monkeypatch, nodes):
    set_node_source(monkeypatch, nodes)
    fr = clipped_spheres_scene().freeze()
    g = lib.Scene(fr)
    rays = random_rays(30000, (-9, -9, -9), (9, 9, 9), 8, finite_fraction=0.1, axis_fraction=0.0)
    check_against_fp64(g.intersect, fr, rays, min_hits=4000)


@pytest.mark.parametrize("name", ["random", "cornell", "week", "pbrt"])
def test_hip_matches_fp64_on_the_benchmark_scenes(scene_cache, name):
    from tests import orc  # ray generation only (camera rays as plain data); not part of the comparison
    sd, cam, fr = scene_cache(name)
    g = lib.Scene(fr)
    wb = g.world_bound()
    lo, hi = np.maximum(wb[:3], -1500), np.minimum(wb[3:], 1500)
    prim = lib.generate_rays(cam(128, 64), lib.render_opts(128, 64, 1), 0, 128 * 64)
    rays = np.concatenate([prim, random_rays(6000, lo, hi, 5, finite_fraction=0.1, axis_fraction=0.0)])
    hits = g.intersect(rays)
    kinds = fr._keep["prims"]["kind"] & 0xFF
    f64 = fp64_ref.Fp64Scene(fr)
    best = f64.closest(rays)
    clear = fp64_ref.clear_rays(f64, rays, best, eps_o=1e-6 * float(np.abs(hi - lo).max()))
    assert clear.mean() > 0.8, clear.mean()
    out = fp64_ref.compare(hits, best, clear, **TOL)
    assert out["n_hits"] > 2000, out
    if f64.n_media:  # scatter points in the fog are among the compared hits
        in_fog = clear & (best["prim"] >= 0) & (kinds[np.maximum(best["prim"], 0)] == abi.RT_MEDIUM)
        assert in_fog.sum() > 300, in_fog.sum()


def test_hip_matches_fp64_from_inside_the_balls():
    """Rays that start inside fog balls take the short cut that skips the tree.  They scatter in
    the fog, or hit the shell from inside, or -- fog without a shell, or a ray that ends first --
    whatever surface lies beyond: the float64 intersector, which knows nothing of balls, must
    agree on all three."""
    from tests.mixed import ball_scene, rays_inside
    sd, balls = ball_scene(seed=11)
    fr = sd.freeze()
    g = lib.Scene(fr)
    assert g.info().n_balls == 3
    rays = rays_inside(balls, 8000, 11)
    hits = g.intersect(rays)
    kinds = fr._keep["prims"]["kind"] & 0xFF
    f64 = fp64_ref.Fp64Scene(fr)
    best = f64.closest(rays)
    clear = fp64_ref.clear_rays(f64, rays, best)
    assert clear.mean() > 0.8, clear.mean()
    out = fp64_ref.compare(hits, best, clear, **TOL)
    assert out["n_hits"] > 6000, out
    won = kinds[np.maximum(best["prim"], 0)][clear & (best["prim"] >= 0)]
    assert (won == abi.RT_MEDIUM).sum() > 2000 and (won != abi.RT_MEDIUM).sum() > 2000  # scatters and ways out
