"""The N > 1 path on CPU: world_size-2 processes over gloo run the same
tile-sharding + reduce(sum) host logic that bench.py runs over RCCL.  The
device renderer is replaced by the oracle (the checker standing in for the GPU),
so this exercises shard ownership, the collective and the bit-exactness of the
assembled frame, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import scenes
from raytracer_amd.distributed import render_frame, tiles_of
from tests import orc

W, H, SPP, DEPTH = 96, 80, 2, 12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _film_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import abi, lib
    sd, camf = scenes.cornell()
    o = orc.OracleScene(sd.freeze())
    cam = camf(W, H)
    filt = lib.make_filter(abi.RT_FILTER_GAUSSIAN, (2.0, 2.0), a=2.0)
    fb = torch.zeros((H, W, 4), dtype=torch.float32)

    def shard(index, count, buf):
        img, st = o.render(cam, orc.render_opts(W, H, SPP, DEPTH, seed=5, shard=(index, count), filter=filt,
                                                flags=abi.RT_RENDER_RAW_FILM), threads=2)
        buf.copy_(torch.from_numpy(img))
        return st

    render_frame(shard, fb, rank, world, dist, resolve=lambda buf: orc.film_resolve(buf.numpy()))
    if rank == 0:
        np.save(out_path, fb.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_filtered_frame_over_two_ranks(tmp_path):
    from raytracer_amd import abi, lib
    out = str(tmp_path / "film.npy")
    mp.spawn(_film_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sd, camf = scenes.cornell()
    filt = lib.make_filter(abi.RT_FILTER_GAUSSIAN, (2.0, 2.0), a=2.0)
    ref, _ = orc.OracleScene(sd.freeze()).render(camf(W, H), orc.render_opts(W, H, SPP, DEPTH, seed=5, filter=filt))
    # partial sums of two ranks are added once more than in the single frame: equal to rounding
    assert np.allclose(np.load(out), ref, rtol=1e-5, atol=1e-6)


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd, camf = scenes.cornell()
    o = orc.OracleScene(sd.freeze())
    cam = camf(W, H)
    fb = torch.zeros((H, W, 4), dtype=torch.float32)

    def shard(index, count, buf):
        img, st = o.render(cam, orc.render_opts(W, H, SPP, DEPTH, seed=5, shard=(index, count)), threads=2)
        buf.copy_(torch.from_numpy(img))
        return st

    st = render_frame(shard, fb, rank, world, dist)
    seg = torch.tensor([float(st.segments)], dtype=torch.float64)
    dist.all_reduce(seg)
    if rank == 0:
        np.save(out_path, np.concatenate([fb.numpy().reshape(-1), seg.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_frame(tmp_path, world):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    data = np.load(out)
    img, seg = data[:-1].reshape(H, W, 4).astype(np.float32), data[-1]
    sd, camf = scenes.cornell()
    ref, st = orc.OracleScene(sd.freeze()).render(camf(W, H), orc.render_opts(W, H, SPP, DEPTH, seed=5))
    assert img.tobytes() == ref.tobytes()  # disjoint tiles: the sum adds every pixel to zeros
    assert seg == st.segments


def test_tile_ownership_is_a_partition():
    for world in (1, 2, 3, 8):
        owned = [tiles_of(1920, 1080, r, world) for r in range(world)]
        flat = sorted(t for o in owned for t in o)
        assert flat == list(range(60 * 34))
        assert max(map(len, owned)) - min(map(len, owned)) <= 1
