"""The N > 1 path with the GPU renderer: two processes over gloo share the test box's one GPU
(RCCL refuses two ranks on one device), each renders its tiles with rt_render_dev, the
framebuffers are summed onto rank 0 and, for a filtered frame, normalised there with
rt_film_resolve_dev.  Unfiltered: the assembled frame is the single-process frame bit for bit.
Filtered (raw films add across ranks with one more rounding per pixel): equal to rounding."""
import os
import socket

import numpy as np
import pytest

W, H, SPP, DEPTH = 160, 96, 4, 20


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, filtered, out_path):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import abi, lib, scenes
    from raytracer_amd.distributed import render_frame
    sd, camf = scenes.next_week(boxes_per_side=8, cube_count=100)
    scene = lib.Scene(sd.freeze(), device=0)
    cam = camf(W, H)
    torch.cuda.set_device(0)
    fb = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    filt = lib.make_filter(abi.RT_FILTER_MITCHELL, (2.0, 2.0), a=1 / 3, b=1 / 3) if filtered else None

    def shard(index, count, buf):
        opts = lib.render_opts(W, H, SPP, DEPTH, seed=5, shard=(index, count), filter=filt,
                               flags=abi.RT_RENDER_RAW_FILM if filtered else 0)
        return scene.render_dev(cam, opts, buf.data_ptr(), stream)

    def reduce_on_host(buf, dst, op):  # gloo carries host tensors
        host = buf.cpu()
        dist.reduce(host, dst=dst, op=op)
        buf.copy_(host)

    class HostReduce:
        ReduceOp = dist.ReduceOp
        reduce = staticmethod(reduce_on_host)

    resolve = (lambda buf: lib.film_resolve_dev(buf.data_ptr(), W, H, 0, stream)) if filtered else None
    st = render_frame(shard, fb, rank, world, HostReduce, resolve=resolve)
    torch.cuda.synchronize()
    assert st.segments > 0
    if rank == 0:
        np.save(out_path, fb.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("filtered", [False, True])
def test_two_gpu_ranks_assemble_the_single_process_frame(tmp_path, filtered):
    import torch.multiprocessing as mp
    from raytracer_amd import abi, lib, scenes
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), filtered, out), nprocs=2, join=True)
    sd, camf = scenes.next_week(boxes_per_side=8, cube_count=100)
    g = lib.Scene(sd.freeze())
    filt = lib.make_filter(abi.RT_FILTER_MITCHELL, (2.0, 2.0), a=1 / 3, b=1 / 3) if filtered else None
    ref, _ = g.render(camf(W, H), lib.render_opts(W, H, SPP, DEPTH, seed=5, filter=filt))
    got = np.load(out)
    if filtered:
        assert np.allclose(got, ref, rtol=1e-5, atol=1e-6)
    else:
        assert got.tobytes() == ref.tobytes()
