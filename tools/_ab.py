import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
CONFIGS = {"C1": ("random", 1200, 800, 16, 50), "C2": ("cornell", 1920, 1080, 256, 50), "C4": ("pbrt", 3840, 2160, 64, 50),
           "C3": ("week", 1920, 1080, 64, 50)}
for name in ("C1", "C4", "C2", "C3"):
    sc, w, h, spp, depth = CONFIGS[name]
    sd, camf = scenes.BUILDERS[sc]()
    g = lib.Scene(sd.freeze(), 4, 0)
    fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    o = lib.render_opts(w, h, spp, depth, seed=0)
    def t(n=3):
        g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            s = g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, s
    for rep in range(3):
        for steal in ("0", "1"):
            os.environ["RT355_LOOP_STEAL"] = steal
            ms, s = t()
            print("%s steal %s: %.2f ms launches %d" % (name, steal, ms, s.trace_launches), flush=True)
    g.close()
