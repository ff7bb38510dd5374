#!/usr/bin/env python3
"""Timeline of one shard's frame: which kernels run, for how long, and the idle gaps between them.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/shard_timeline.py run <shard> <count>
                 python3 tools/shard_timeline.py report DIR
`run` renders shard <shard> of <count> of the headline frame three times; `report` prints the
kernels of the last frame in the trace (start offset, duration, gap to the previous kernel)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(index, count):
    import torch
    from raytracer_amd import lib, scenes
    sd, camf = scenes.BUILDERS["week"]()
    g = lib.Scene(sd.freeze(), 4, 0)
    w, h = 1920, 1080
    fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    o = lib.render_opts(w, h, 64, 50, seed=0, shard=(index, count))
    for _ in range(3):
        g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()


def report(d):
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "k_generate" in r[2]]
    frame = rows[starts[-1]:]
    while frame and "k_resolve" not in frame[-1][2]:
        frame.pop()
    t0, prev = frame[0][0], frame[0][0]
    busy = 0
    for s, e, name in frame:
        short = name.split("(")[0].replace("void rt::", "")[:48]
        print("%9.3f ms  %8.3f ms  gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev) / 1e6, short))
        busy += e - s
        prev = e
    print("frame %.3f ms, kernels %.3f ms, idle %.3f ms" % ((frame[-1][1] - t0) / 1e6, busy / 1e6,
                                                          (frame[-1][1] - t0 - busy) / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]))
    else:
        report(sys.argv[2])
