#!/usr/bin/env python3
"""At which bounce should the one-launch kernel take over?  Shard 0 of 8, 4 and 2 of the headline
frame with the library's own policy and with the switch fixed at bounce 1 ... 24 (RT355_LOOP_FROM).
usage (GPU box): loop_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
sd, camf = scenes.BUILDERS["week"]()
fr = sd.freeze()
def scene():  # the hooks are read when a scene is placed on the device
    g = lib.Scene(fr, 4, 0)
    g.upload()
    return g
g = scene()
w, h = 1920, 1080
fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def t(o, n=4):
    g.render_dev(camf(w, h), o, fb.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        s = g.render_dev(camf(w, h), o, fb.data_ptr(), st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, s
for n in (1, 8, 4, 2):
    o = lib.render_opts(w, h, 64, 50, seed=0, shard=(0, n))
    os.environ.pop("RT355_LOOP_FROM", None); os.environ.pop("RT355_LOOP_BELOW", None)
    g = scene()
    ms, s = t(o)
    print("shards %d auto: %.2f ms launches %d rays %s" % (n, ms, s.trace_launches, list(s.bounce_rays[:10])), flush=True)
    for f in (1, 2, 3, 4, 6, 8, 12, 16, 24):
        os.environ["RT355_LOOP_FROM"] = str(f)
        g = scene()
        ms, s = t(o)
        print("shards %d from %d: %.2f ms" % (n, f, ms), flush=True)
    os.environ.pop("RT355_LOOP_FROM", None)
