import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
sd, camf = scenes.BUILDERS["week"]()
g = lib.Scene(sd.freeze(), 4, 0)
w, h = 1920, 1080
fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "ho")
for n in (8, 1):
    o = lib.render_opts(w, h, 64, 50, seed=0, shard=(0, n), flags=abi.RT_RENDER_COUNTERS)
    for rep in range(2):
        os.environ["RT355_TAIL_FILE"] = os.path.join(out, "waves_%d.txt" % n)
        g.render_dev(camf(w, h), o, fb.data_ptr(), st)
        torch.cuda.synchronize()
