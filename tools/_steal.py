import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
sd, camf = scenes.BUILDERS["week"]()
g = lib.Scene(sd.freeze(), 4, 0)
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
for n in (8, 4, 2, 1):
    o = lib.render_opts(w, h, 64, 50, seed=0, shard=(0, n))
    sums = set()
    for rep in range(2):
        for steal in ("0", "1"):
            os.environ["RT355_LOOP_STEAL"] = steal
            ms, s = t(o)
            sums.add(hashlib.sha256(fb.cpu().numpy().tobytes()).hexdigest())
            print("shards %d steal %s: %.2f ms segments %d launches %d" % (n, steal, ms, s.segments, s.trace_launches), flush=True)
    print("shards %d distinct images %d" % (n, len(sums)), flush=True)
