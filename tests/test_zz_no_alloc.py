"""rt_intersect_dev_noalloc (the entry point that allocates nothing) against
rt_intersect_dev, through the C ABI: same hit records bit for bit on two calls in a
row, the second with a smaller input, for trees read from LDS and from global memory;
its argument checks; and that the oracle agrees with both."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, lib
from tests import orc
from tests.conftest import random_rays, set_node_source

pytestmark = pytest.mark.gpu


def _device_hits(g, entry, d_rays, n, stream):
    import torch
    d_hits = torch.full((max(n, 1), 8), -7.0, dtype=torch.float32, device="cuda")
    entry(d_rays.data_ptr(), n, d_hits.data_ptr(), stream)
    torch.cuda.synchronize()
    return d_hits.cpu().numpy()[:n].view(abi.hit_dtype).reshape(-1)


@pytest.mark.parametrize("nodes", ["lds", "global", "quads"])
def test_no_alloc_entry_point_matches_intersect_dev(monkeypatch, scene_cache, nodes):
    import torch
    set_node_source(monkeypatch, nodes)
    _, cam, fr = scene_cache("week")
    g, o = lib.Scene(fr), orc.OracleScene(fr)
    wb = o.world_bound()
    lo, hi = np.maximum(wb[:3], -2000), np.minimum(wb[3:], 2000)
    stream = torch.cuda.current_stream().cuda_stream
    L = lib.load()

    rays = random_rays(100_003, lo, hi, 21)
    d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
    # not resident yet: the entry point must refuse rather than upload (which allocates)
    rc = L.rt_intersect_dev_noalloc(g._h, d_rays.data_ptr(), len(rays), d_rays.data_ptr(), 0, stream)
    assert rc == abi.RT_ERR_INVALID and b"rt_scene_upload" in L.rt_last_error()
    g.upload()
    # two calls in a row, the second with a smaller input (a prefix of other rays)
    for n, seed in ((100_003, 21), (4_097, 22)):
        rays = random_rays(n, lo, hi, seed)
        d_rays = torch.from_numpy(rays.view(np.float32).reshape(-1, 8)).cuda()
        want = _device_hits(g, g.intersect_dev, d_rays, n, stream)
        got = _device_hits(g, g.intersect_dev_noalloc, d_rays, n, stream)
        # ... and it really allocates nothing: free device memory is the same before and after two
        # calls on buffers that exist already (the runtime reports whole bytes of the device)
        d_hits = torch.empty((n, 8), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        for _ in range(2):
            g.intersect_dev_noalloc(d_rays.data_ptr(), n, d_hits.data_ptr(), stream)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, (free_before, free_after)
        assert d_hits.cpu().numpy().view(abi.hit_dtype).reshape(-1).tobytes() == want.tobytes()
        assert (want["prim"] != abi.RT_MISS).mean() > 0.2
        assert got.tobytes() == want.tobytes(), "%s n=%d" % (nodes, n)
        assert got.tobytes() == o.intersect(rays).tobytes()
    # empty batch, oversize batch, null buffers
    assert L.rt_intersect_dev_noalloc(g._h, None, 0, None, 0, stream) == abi.RT_OK
    rc = L.rt_intersect_dev_noalloc(g._h, d_rays.data_ptr(), abi.RT_NOALLOC_MAX_RAYS + 1, d_rays.data_ptr(), 0, stream)
    assert rc == abi.RT_ERR_LIMIT and b"at most" in L.rt_last_error()
    assert L.rt_intersect_dev_noalloc(g._h, None, 5, None, 0, stream) == abi.RT_ERR_INVALID
    g.close()
