#!/usr/bin/env python3
"""Builds variants/librt355_<tag>.so for A/B runs.  usage: build_variants.py tag=-DFOO=1,-DBAR=2 ... (tag 'base' = no flags)"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import build  # noqa: E402

os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)


def one(arg):
    tag, _, flags = arg.partition("=")
    extra = [f for f in flags.split(",") if f]
    # a marker flag keeps the object directory of the plain variant apart from the shipped one
    out = build.build(extra=extra or ["-DRT_VARIANT_BASE"], lib=os.path.join(ROOT, "variants", "librt355_%s.so" % tag))
    return tag, out


with ThreadPoolExecutor(max_workers=2) as pool:
    for tag, out in pool.map(one, sys.argv[1:]):
        print(tag, out)
