#!/usr/bin/env python3
"""Per-wave lifetimes of the one-launch kernel (needs a library built with -DRT_TAIL_PROBE, loaded through
RT355_LIB): renders shard `index` of `count` of the headline frame once with counters and summarises the
records the kernel left (when its queue ran dry, when it ended, what it still held, turns and batches after).
usage: tail_probe.py [index count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RT355_TAIL_FILE"] = "/tmp/rt355_tail.txt"
import torch  # noqa: E402,F401
from raytracer_amd import abi, lib, scenes  # noqa: E402

index, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 1)
sd, camf = scenes.next_week()
scene = lib.Scene(sd.freeze(), 4, abi.RT_SPLIT_SAH)
w, h = 1920, 1080
fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
for _ in range(2):
    scene.render_dev(camf(w, h), lib.render_opts(w, h, 64, 50, seed=0, shard=(index, count), flags=abi.RT_RENDER_COUNTERS),
                     fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
r = np.loadtxt("/tmp/rt355_tail.txt")
start, dry, end, held, turns, batches, fill = r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
print("waves %d; launch ends %.3f ms" % (len(r), end.max()))
ok = dry >= 0
print("queue dry: min %.3f median %.3f max %.3f ms" % (dry[ok].min(), np.median(dry[ok]), dry[ok].max()))
print("wave end: 10%% %.3f median %.3f 90%% %.3f 99%% %.3f max %.3f ms" % tuple(np.percentile(end, [10, 50, 90, 99, 100])))
life = end[ok] - dry[ok]
print("after dry: median %.3f 90%% %.3f max %.3f ms; held then: median %d max %d; turns after: median %d max %d; "
      "batches after: median %d max %d, mean fill %.1f" % (np.median(life), np.percentile(life, 90), life.max(), np.median(held[ok]),
                                                           held[ok].max(), np.median(turns[ok]), turns[ok].max(), np.median(batches[ok]),
                                                           batches[ok].max(), fill[ok].sum() / max(batches[ok].sum(), 1)))
late = np.argsort(end)[-5:]
for k in late:
    print("late wave %d: dry %.3f end %.3f held %d turns %d batches %d fill %.1f" % (r[k, 0], dry[k], end[k], held[k], turns[k], batches[k],
                                                                                   fill[k] / max(batches[k], 1)))
