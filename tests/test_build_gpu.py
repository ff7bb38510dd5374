"""HLBVH built on the GPU (Morton codes, radix sort, treelets; rt_build_gpu.hip) against
the host builder and the oracle: the same nodes and primitive order, bit for bit."""
import numpy as np
import pytest

from raytracer_amd import abi, lib
from raytracer_amd.scene import SceneData
from tests import orc
from tests.mixed import mixed_scene


def big_scene(n_spheres, n_tris, seed):
    s = SceneData()
    s.lambertian((1, 1, 1))
    rng = np.random.default_rng(seed)
    s.add_sphere_block(rng.random((n_spheres, 3)) * 100, rng.random(n_spheres) * 0.3 + 0.05)
    if n_tris:
        s.add_triangles(rng.random((n_tris, 1, 3)) * 100 + (rng.random((n_tris, 3, 3)) - 0.5))
    return s.freeze()


@pytest.mark.gpu
@pytest.mark.parametrize("max_prims", [1, 4, 255])
def test_device_hlbvh_equals_host(max_prims, scene_cache):
    cases = [mixed_scene(seed=3, n_each=60).freeze(), scene_cache("week")[2], scene_cache("cornell")[2],
             big_scene(150000, 50000, 5)]
    # many primitives with the same Morton code (one treelet, leaves decided by the count alone)
    s = SceneData()
    s.lambertian((1, 1, 1))
    s.add_sphere_block(np.tile(np.float32([[1, 2, 3]]), (300, 1)), 0.5)
    s.add_sphere((50, 50, 50), 1.0)
    cases.append(s.freeze())
    for fr in cases:
        host = lib.Scene(fr, max_prims, abi.RT_SPLIT_HLBVH)
        dev = lib.Scene(fr, max_prims, abi.RT_SPLIT_HLBVH, build_on_device=True)
        assert dev.nodes().tobytes() == host.nodes().tobytes()
        assert (dev.prim_order() == host.prim_order()).all()
        assert dev.info().max_depth == host.info().max_depth
    o = orc.OracleScene(cases[0], max_prims, abi.RT_SPLIT_HLBVH)
    dev = lib.Scene(cases[0], max_prims, abi.RT_SPLIT_HLBVH, build_on_device=True)
    assert dev.nodes().tobytes() == o.nodes().tobytes()


@pytest.mark.gpu
def test_device_build_needs_hlbvh_and_a_device():
    fr = mixed_scene(seed=3, n_each=5).freeze()
    with pytest.raises(lib.RtError) as e:
        lib.Scene(fr, 4, abi.RT_SPLIT_SAH, build_on_device=True)
    assert e.value.code == abi.RT_ERR_INVALID
    with pytest.raises(lib.RtError) as e:
        lib.Scene(fr, 4, abi.RT_SPLIT_HLBVH, device=99, build_on_device=True)
    assert e.value.code == abi.RT_ERR_NO_DEVICE
