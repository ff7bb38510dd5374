"""Frames enqueued without waiting (rt_render_dev with stats == NULL) against waited-for ones: same image.
usage: async_check.py [width height]; run two at once to see two processes share one GPU."""
import faulthandler, sys, os
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, lib, scenes
sd, camf = scenes.next_week()
g = lib.Scene(sd.freeze(), 4, 0)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 270)
fb = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
fb2 = torch.zeros_like(fb)
st = torch.cuda.current_stream().cuda_stream
o = lib.render_opts(w, h, 2 if w > 1000 else 4, 50, seed=0)
s = g.render_dev(camf(w, h), o, fb.data_ptr(), st)
print("sync frame", s.segments, s.trace_launches, flush=True)
for i in range(3):
    g.render_dev(camf(w, h), o, fb2.data_ptr(), st, want_stats=False)
    print("enqueued", i, flush=True)
torch.cuda.synchronize()
print("same image:", bool((fb == fb2).all()), flush=True)
s = g.render_dev(camf(w, h), o, fb2.data_ptr(), st)
print("sync again", s.segments, flush=True)
