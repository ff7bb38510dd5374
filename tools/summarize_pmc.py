#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel. usage: summarize_pmc.py <dir> [kernel substring]"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else "k_bounce"
tot = collections.defaultdict(float)
calls = collections.defaultdict(int)
for f in sorted(glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        if match not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        seen.add(r["Dispatch_Id"])
    for c in {r2 for r2 in tot}:
        pass
    calls[f] = len(seen)
out = {"kernel": match, "dispatches_per_pass": sorted(set(calls.values())), "counters": dict(sorted(tot.items()))}
c = tot
d = {}
if c.get("SQ_WAVE_CYCLES"):
    d["wait_any_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    d["wait_inst_any_frac"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["active_inst_any_frac"] = c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]
    d["active_inst_valu_frac"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    d["mean_waves_per_busy_cycle"] = c["SQ_WAVE_CYCLES"] / max(c.get("SQ_BUSY_CYCLES", 1), 1)
if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
    # lanes active per VALU instruction cycle: 64 = no divergence
    d["valu_active_lane_frac"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
    d["l1_miss_to_l2_frac"] = c.get("TCP_TCC_READ_REQ_sum", 0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "FETCH_SIZE" in c:
    # rocprofv3 reports KiB; on gfx950 a wide coalesced read stream shows half its bytes
    d["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
    d["fetch_bytes_corrected_x2"] = c["FETCH_SIZE"] * 2048
if "WRITE_SIZE" in c:
    d["write_bytes"] = c["WRITE_SIZE"] * 1024
out["derived"] = d
print(json.dumps(out, indent=1))
