#!/usr/bin/env python3
"""Measured HBM rates of this GPU (denominators quoted beside the 8 TB/s data-sheet
figure): device-to-device copy (read + write) and a read-only reduction."""
import json

import torch

n = 1 << 30  # 4 GiB of float32 per buffer: far larger than the 256 MiB Infinity Cache
src = torch.empty(n, dtype=torch.float32, device="cuda").fill_(1.0)
dst = torch.empty_like(src)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"device": torch.cuda.get_device_name(0), "buffer_GiB": n * 4 / 2**30}
ms = timed(lambda: dst.copy_(src))
out["copy_GBs"] = round(2 * n * 4 / ms / 1e6, 1)
ms = timed(lambda: src.sum())
out["read_GBs"] = round(n * 4 / ms / 1e6, 1)
ms = timed(lambda: dst.fill_(0.0))
out["write_GBs"] = round(n * 4 / ms / 1e6, 1)
p = torch.cuda.get_device_properties(0)
out.update(cus=p.multi_processor_count, hbm_GiB=round(p.total_memory / 2**30, 1),
           l2_MiB=p.L2_cache_size / 2**20)
print(json.dumps(out))
