#!/usr/bin/env python3
"""The large synthetic scene split over 1, 2, 4 and 8 ranks, each rank's shard rendered alone on one
GPU (what strong scaling can be expected before any multi-GPU run): the first, a middle and the last
shard of every rank count at `spp` samples per pixel.  usage: shard_times_c5.py [spp, default 128]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sd, camf = scenes.BUILDERS["synthetic"]()
g = lib.Scene(sd.freeze(), 4, abi.RT_SPLIT_SAH)
w, h = 3840, 2160
fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
whole = None
for n in (1, 2, 4, 8):
    worst, segs = 0.0, 0
    for k in sorted({0, n // 2, n - 1}):
        o = lib.render_opts(w, h, spp, 8, seed=0, shard=(k, n))
        g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        worst = max(worst, dt)
        segs = max(segs, s.segments)
    if n == 1:
        whole = worst
    print("%d shards: slowest of the shards tried %.1f ms/frame (%.1f if the frame divided evenly), %d segments" %
          (n, worst, whole / n, segs), flush=True)
