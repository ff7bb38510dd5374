#!/usr/bin/env python3
"""What strong scaling can be expected before any multi-GPU run: every shard of the headline
frame for 1, 2, 4 and 8 ranks rendered alone on one GPU; the slowest shard bounds the frame.
usage: shard_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
sd, camf = scenes.BUILDERS["week"]()
g = lib.Scene(sd.freeze(), 4, 0)
w, h = 1920, 1080
fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    worst = 0.0
    for k in range(n):
        o = lib.render_opts(w, h, 64, 50, seed=0, shard=(k, n))
        g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            s = g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3 * 1e3
        worst = max(worst, dt)
    if n == 1:
        whole = worst
    print("%d shards: slowest shard %.2f ms/frame, %.2f if the frame divided evenly" % (n, worst, whole / n))
