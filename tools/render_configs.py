#!/usr/bin/env python3
"""Renders the five benchmark configurations at full size on one GPU and prints one JSON
line each: Mrays/s, ms/frame, traversal counters, algorithmic bytes, and a
bit-exactness spot check of sampled pixels against the CPU oracle.
usage: render_configs.py [C1|C2|C3|C4|C5 ...] [--spp-scale F] [--check N]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

from raytracer_amd import abi, lib, scenes

CONFIGS = {
    "C1": ("random", 1200, 800, 16, 50, {}),
    "C2": ("cornell", 1920, 1080, 256, 50, {}),
    "C3": ("week", 1920, 1080, 64, 50, {}),
    "C4": ("pbrt", 3840, 2160, 64, 50, {}),
    "C5": ("synthetic", 3840, 2160, 128, 8, {}),
}
PRIM_BYTES = [16, 32, 36, 24, 24, 124, 32, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C1", "C2", "C3", "C4"])
    ap.add_argument("--spp-scale", type=float, default=1.0)
    ap.add_argument("--check", type=int, default=64, help="pixels compared with the oracle")
    ap.add_argument("--split", type=int, default=abi.RT_SPLIT_SAH)
    ap.add_argument("--sampler", default="random", choices=["random", "halton", "sobol", "zero_two", "stratified"])
    args = ap.parse_args()
    sampler = ["random", "halton", "sobol", "zero_two", "stratified"].index(args.sampler)
    for name in args.configs:
        scene_name, w, h, spp, depth, kw = CONFIGS[name]
        spp = max(1, int(round(spp * args.spp_scale)))
        # stratified: the squarest grid with spp cells
        skw = dict(sampler=sampler, sampler_arg=max(d for d in range(1, int(spp ** 0.5) + 1) if spp % d == 0)
                   if args.sampler == "stratified" else 0)
        t0 = time.time()
        sd, camf = scenes.BUILDERS[scene_name](**kw)
        fr = sd.freeze()
        g = lib.Scene(fr, 4, args.split)
        g.upload()
        info = g.info()
        setup_s = time.time() - t0
        cam = camf(w, h)
        img, cst = g.render(cam, lib.render_opts(w, h, spp, depth, seed=0, flags=abi.RT_RENDER_COUNTERS, **skw))
        img2, st = g.render(cam, lib.render_opts(w, h, spp, depth, seed=0, flags=abi.RT_RENDER_TIMING, **skw))
        assert img.tobytes() == img2.tobytes()
        alg = cst.segments * 64 + cst.nodes_visited * 32 + sum(
            cst.prims_tested[k] * PRIM_BYTES[k] for k in range(8))
        out = {"config": name, "sampler": args.sampler, "scene": scene_name, "width": w, "height": h, "spp": spp, "depth": depth,
               "prims": info.n_prims, "nodes": info.n_nodes, "tree_depth": info.max_depth,
               "build_ms": round(info.build_ms, 1), "setup_s": round(setup_s, 1),
               "device_MB": round(info.device_bytes / 1e6, 1),
               "paths": st.paths, "segments": st.segments, "ms_per_frame": round(st.total_ms, 2),
               "Mrays_s": round(st.segments / st.total_ms / 1e3, 1),
               "trace_ms": round(st.trace_ms, 2), "launches": st.trace_launches, "batches": st.batches,
               "nodes_per_ray": round(cst.nodes_visited / cst.segments, 2),
               "prims_per_ray": round(sum(cst.prims_tested) / cst.segments, 2),
               "alg_bytes_per_ray": round(alg / cst.segments, 1),
               "alg_GBs": round(alg / (st.trace_ms * 1e-3) / 1e9, 1),
               "mean_rgb": [round(float(v), 5) for v in img[..., :3].mean((0, 1))],
               "finite": bool(np.isfinite(img).all())}
        if args.check:
            from tests import spot_check  # the oracle stays behind tests/
            bad = spot_check.pixels_different(fr, 4, args.split, cam, w, h, spp, depth, img, args.check, **skw)
            out["oracle_pixels_checked"] = args.check
            out["oracle_pixels_different"] = bad
        print(json.dumps(out), flush=True)
        g.close()


if __name__ == "__main__":
    main()
