#!/usr/bin/env python3
"""Micro-benchmark of rt_intersect_dev / rt_occluded_dev: primary rays (coherent)
and cosine-distributed bounce rays (incoherent) on one scene, rays resident in
HBM.  usage: trace_bench.py [scene] [log2 rays] [scene kwargs as k=v] [--orders]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from raytracer_amd import abi, lib, scenes


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "week"
    log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[3:] if "=" in a)}
    sd, camf = scenes.BUILDERS[name](**kw)
    fr = sd.freeze()
    scene = lib.Scene(fr)
    scene.upload()
    info = scene.info()
    n = 1 << log2n
    w = 4096
    h = min(n // w, 4096)
    spp = n // (w * h)  # beyond 2^24 rays: more samples per pixel of a 4096 x 4096 frame
    cam = camf(w, h)
    opts = lib.render_opts(w, h, spp, seed=0)
    L = lib.load()
    import ctypes as C
    d_rays = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_hits = torch.empty((n, 8), dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    # primary rays: generated on the device in pieces through the host API (setup only)
    piece = 1 << 22
    for at in range(0, n, piece):
        r = lib.generate_rays(cam, opts, at, min(piece, n - at))
        d_rays[at:at + piece] = torch.from_numpy(r.view(np.float32).reshape(-1, 8)).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    out = {"scene": name, "prims": info.n_prims, "nodes": info.n_nodes, "rays": n}
    ms = timed(lambda: scene.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), stream))
    cnt = scene.intersect_count(d_rays.data_ptr(), n, d_hits.data_ptr())
    out["primary"] = {"ms": round(ms, 3), "Mrays_s": round(n / ms / 1e3, 1),
                      "nodes_per_ray": round(float(cnt[0]) / n, 2),
                      "prims_per_ray": round(float(cnt[1:].sum()) / n, 2)}
    ms = timed(lambda: scene.occluded_dev(d_rays.data_ptr(), n, d_occ.data_ptr(), stream))
    out["primary_any_hit"] = {"ms": round(ms, 3), "Mrays_s": round(n / ms / 1e3, 1)}
    # bounce rays: from the primary hit points, direction = normal + uniform unit vector
    hits = d_hits.clone()
    t = hits[:, 0:1]
    prim = hits[:, 1].view(torch.int32)
    nrm = hits[:, 4:7]
    o, d = d_rays[:, 0:3], d_rays[:, 3:6]
    hit = (prim != -1)
    g = torch.Generator(device="cuda").manual_seed(1)
    v = torch.randn((n, 3), device="cuda", generator=g)
    v = v / v.norm(dim=1, keepdim=True)
    nf = torch.where((nrm * d).sum(1, keepdim=True) < 0, nrm, -nrm)
    nd = nf + v
    nd = nd / nd.norm(dim=1, keepdim=True).clamp_min(1e-20)
    p = o + t * d + nf * 1e-3 * (1.0 + (o + t * d).abs().amax(1, keepdim=True))
    b_rays = d_rays.clone()
    b_rays[:, 0:3] = torch.where(hit[:, None], p, o)
    b_rays[:, 3:6] = torch.where(hit[:, None], nd, d)
    b_rays = b_rays[hit].contiguous()
    m = b_rays.shape[0]
    ms = timed(lambda: scene.intersect_dev(b_rays.data_ptr(), m, d_hits.data_ptr(), stream))
    cnt = scene.intersect_count(b_rays.data_ptr(), m, d_hits.data_ptr())
    out["bounce1"] = {"rays": m, "ms": round(ms, 3), "Mrays_s": round(m / ms / 1e3, 1),
                      "nodes_per_ray": round(float(cnt[0]) / m, 2),
                      "prims_per_ray": round(float(cnt[1:].sum()) / m, 2)}
    ms = timed(lambda: scene.occluded_dev(b_rays.data_ptr(), m, d_occ.data_ptr(), stream))
    out["bounce1_any_hit"] = {"ms": round(ms, 3), "Mrays_s": round(m / ms / 1e3, 1)}
    # the same bounce rays in random order and sorted by the Morton code of their origin: what
    # the order of a ray array is worth
    if "--orders" in sys.argv:
        perm = torch.randperm(m, device="cuda", generator=g)
        shuffled = b_rays[perm].contiguous()
        ms = timed(lambda: scene.intersect_dev(shuffled.data_ptr(), m, d_hits.data_ptr(), stream))
        out["bounce1_shuffled"] = {"ms": round(ms, 3), "Mrays_s": round(m / ms / 1e3, 1)}
        lo, hi = b_rays[:, 0:3].amin(0), b_rays[:, 0:3].amax(0)
        q = ((b_rays[:, 0:3] - lo) / (hi - lo).clamp_min(1e-20) * 1023.0).to(torch.int64).clamp_(0, 1023)

        def spread(x):
            x = (x | (x << 16)) & 0x030000FF
            x = (x | (x << 8)) & 0x0300F00F
            x = (x | (x << 4)) & 0x030C30C3
            return (x | (x << 2)) & 0x09249249
        code = (spread(q[:, 2]) << 2) | (spread(q[:, 1]) << 1) | spread(q[:, 0])
        ordered = b_rays[torch.argsort(code)].contiguous()
        ms = timed(lambda: scene.intersect_dev(ordered.data_ptr(), m, d_hits.data_ptr(), stream))
        out["bounce1_origin_sorted"] = {"ms": round(ms, 3), "Mrays_s": round(m / ms / 1e3, 1)}
        # coarse cells only (32 per axis), random order inside a cell
        q5 = q >> 5
        code5 = (spread(q5[:, 2]) << 2) | (spread(q5[:, 1]) << 1) | spread(q5[:, 0])
        coarse = shuffled[torch.argsort(code5[perm], stable=True)].contiguous()
        ms = timed(lambda: scene.intersect_dev(coarse.data_ptr(), m, d_hits.data_ptr(), stream))
        out["bounce1_cell32_sorted"] = {"ms": round(ms, 3), "Mrays_s": round(m / ms / 1e3, 1)}
        # origin cells of 128 per axis
        q7 = q >> 3
        code7 = (spread(q7[:, 2]) << 2) | (spread(q7[:, 1]) << 1) | spread(q7[:, 0])
        mid = shuffled[torch.argsort(code7[perm], stable=True)].contiguous()
        ms = timed(lambda: scene.intersect_dev(mid.data_ptr(), m, d_hits.data_ptr(), stream))
        out["bounce1_cell128_sorted"] = {"ms": round(ms, 3), "Mrays_s": round(m / ms / 1e3, 1)}
        del shuffled, ordered, coarse, mid
    # What would it be worth if each XCD traced one part of the scene only, so that the eight L2s
    # hold eight different parts of the tree?  Wave w of the global-memory kernels claims the ray
    # blocks b with b % 32 == w % 32 and sits on XCD (w / 4) % 8, so blocks 4x ... 4x + 3 of every 32
    # go to XCD x: the array is re-ordered so that those blocks hold rays of part x (vertical strips
    # of the image, or slabs of the scene along x), against the same rays in their original order.
    if "--xcd" in sys.argv:
        pix = torch.arange(n, device="cuda")[hit]  # the pixel (path) each bounce ray came from
        px = (pix // spp) % w

        def interleave(part):
            order = torch.argsort(part, stable=True)
            counts = torch.bincount(part, minlength=8)
            keep = int(counts.min().item()) // 256 * 256
            starts = torch.cumsum(counts, 0) - counts
            per = [order[int(starts[x]):int(starts[x]) + keep].view(-1, 256) for x in range(8)]
            return torch.stack(per, dim=1).reshape(-1)

        for label, part in (("strips", px * 8 // w),
                            ("slabs", ((b_rays[:, 0] - b_rays[:, 0].min()) /
                                       (b_rays[:, 0].max() - b_rays[:, 0].min() + 1e-9) * 8).long().clamp_(0, 7))):
            sel = interleave(part)
            by_part = b_rays[sel].contiguous()
            same = b_rays[torch.sort(sel).values].contiguous()
            k = by_part.shape[0]
            ms_a = timed(lambda: scene.intersect_dev(by_part.data_ptr(), k, d_hits.data_ptr(), stream))
            ms_b = timed(lambda: scene.intersect_dev(same.data_ptr(), k, d_hits.data_ptr(), stream))
            out["bounce1_xcd_" + label] = {"rays": k, "Mrays_s": round(k / ms_a / 1e3, 1),
                                           "same_rays_in_order_Mrays_s": round(k / ms_b / 1e3, 1)}
            del by_part, same
    # host-pointer entry points: PCIe staging included (rays up, hits down)
    import time
    h_rays = b_rays[:1 << 22].cpu().numpy().view(abi.ray_dtype).reshape(-1)
    scene.intersect(h_rays)
    t0 = time.perf_counter()
    scene.intersect(h_rays)
    dt = time.perf_counter() - t0
    out["bounce1_host_pointers"] = {"rays": len(h_rays), "ms": round(dt * 1e3, 3),
                                    "Mrays_s": round(len(h_rays) / dt / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
