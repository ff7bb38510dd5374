#!/usr/bin/env python3
"""Build times of the 11 M primitive scene: host SAH, host HLBVH, device HLBVH."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import abi, lib, scenes

sd, cam = scenes.synthetic()
fr = sd.freeze()
out = {"prims": sd.n_prims()}
ref = None
for name, split, dev in (("host_sah", abi.RT_SPLIT_SAH, False), ("host_hlbvh", abi.RT_SPLIT_HLBVH, False),
                         ("device_hlbvh", abi.RT_SPLIT_HLBVH, True), ("device_hlbvh_again", abi.RT_SPLIT_HLBVH, True)):
    t0 = time.time()
    s = lib.Scene(fr, 4, split, build_on_device=dev)
    info = s.info()
    out[name] = {"create_s": round(time.time() - t0, 3), "build_ms": round(info.build_ms, 1), "nodes": info.n_nodes}
    if split == abi.RT_SPLIT_HLBVH:
        nodes = s.nodes().tobytes()
        if ref is None:
            ref = nodes
        out[name]["equals_host"] = nodes == ref
    s.close()
print(json.dumps(out))
