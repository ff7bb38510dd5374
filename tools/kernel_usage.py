#!/usr/bin/env python3
"""Registers, spills, scratch and occupancy of every kernel of librt355, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (compiles rt_kernels.hip to a throw-away object).
usage: tools/kernel_usage.py [substring of kernel names] [extra -D flags...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import build  # noqa: E402

want = [a for a in sys.argv[1:] if not a.startswith("-")]
extra = [a for a in sys.argv[1:] if a.startswith("-")]
with tempfile.TemporaryDirectory() as tmp:
    cmd = ["/opt/rocm/bin/hipcc", *build.FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-I",
           os.path.join(ROOT, "include"), "-I", build.CSRC, "-x", "hip", "-c",
           os.path.join(build.CSRC, "rt_kernels.hip"), "-o", os.path.join(tmp, "k.o")]
    text = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    body = m.group(1)
    if body.startswith("Function Name:"):
        name = subprocess.run(["c++filt", body.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
keys = ["VGPRs", "AGPRs", "SGPRs", "VGPR Spill", "SGPR Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
        "LDS Size [bytes/block]"]
print("%-44s %s" % ("kernel", "  ".join(k.split(" [")[0] for k in keys)))
for r in rows:
    if want and not any(w in r["name"] for w in want):
        continue
    print("%-44s %s" % (r["name"][-44:], "  ".join("%6s" % r.get(k, "?") for k in keys)))
